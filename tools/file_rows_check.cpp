#include "../torbi_amd/csrc/file_rows.hpp"
#include <cstdio>
#include <cstdlib>
#include <string>
#include <vector>
#include <fcntl.h>
int main() {
    // write 40 files natively, read them back into padded rows, compare
    const int n = 40;
    std::vector<std::string> names; std::vector<std::vector<char>> bodies;
    std::vector<const char*> paths; std::vector<const void*> data; std::vector<int64_t> bytes;
    for (int k = 0; k < n; ++k) {
        names.push_back("/tmp/torbi_file_rows_check_" + std::to_string(k) + ".bin");
        bodies.emplace_back((size_t)(1000 + 977 * k));
        for (size_t i = 0; i < bodies[k].size(); ++i) bodies[k][i] = (char)(i * 31 + k);
    }
    for (int k = 0; k < n; ++k) { paths.push_back(names[k].c_str()); data.push_back(bodies[k].data()); bytes.push_back((int64_t)bodies[k].size()); }
    int err = 0;
    if (filerows::write_files(paths.data(), data.data(), bytes.data(), n, 5, &err)) { printf("write failed %d\n", err); return 1; }
    std::vector<int> fds; std::vector<int64_t> offs, zeros; std::vector<std::vector<char>> rows; std::vector<void*> dst;
    for (int k = 0; k < n; ++k) {
        fds.push_back(open(names[k].c_str(), O_RDONLY)); offs.push_back(7); bytes[k] -= 7; zeros.push_back(100 + k);
        rows.emplace_back((size_t)(bytes[k] + zeros[k]), (char)0x55); dst.push_back(rows[k].data());
    }
    if (filerows::read_rows(fds.data(), offs.data(), bytes.data(), dst.data(), zeros.data(), n, 7, &err)) { printf("read failed %d\n", err); return 1; }
    for (int k = 0; k < n; ++k) {
        for (int64_t i = 0; i < bytes[k]; ++i) if (rows[k][i] != bodies[k][i + 7]) { printf("mismatch %d\n", k); return 1; }
        for (int64_t i = 0; i < zeros[k]; ++i) if (rows[k][bytes[k] + i] != 0) { printf("padding %d\n", k); return 1; }
        close(fds[k]);
    }
    // a short file is reported
    bytes[3] += 1000;
    fds[3] = open(names[3].c_str(), O_RDONLY);
    std::vector<char> big((size_t)(bytes[3] + zeros[3])); dst[3] = big.data();
    const int rc = filerows::read_rows(fds.data() + 3, offs.data() + 3, bytes.data() + 3, dst.data() + 3, zeros.data() + 3, 1, 2, &err);
    printf("short read rc %d err %d\nok\n", rc, err);
    return rc == -1 ? 0 : 1;
}
