"""Headline benchmark: timesteps decoded / second, 1440 states, batch 512 (BASELINE.json).

    python bench.py [--gpus N] [--steps K] [--warmup W]                  (N > 1: starts its N ranks itself)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W
    python bench.py --workload c4 [--gpus N] [--files 40000]             (BASELINE configs[3], strong scaling)

A "step" is ONE decode (forward recurrence + final argmax + backtrace) of one synthetic batch of 512
sequences x 500 frames x 1440 states (BASELINE configs[2]) that is already resident in HBM.  Consecutive
steps are decoded in groups (torbi_amd.DecodePipeline(group=8)): the batches of a group share ONE forward
launch of the time-resident kernel and one backtrace launch; groups alternate between two HIP streams.
With N GPUs every rank decodes its own batches (weak scaling, batch items are independent) and the decoded
indices of every batch are all-gathered over RCCL inside the timed region.
Rank 0 prints one JSON line; see DESIGN.md "Measurement" for every field.  Without a HIP device the same
launch / rendezvous / planning code runs over gloo and the line carries "dry_run": true and no value.
"""
import argparse
import json
import math
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0                 # MI355X_MICROARCH.md: 8.0 TB/s spec
PCIE_PEAK_GBS = 63.0                  # MI355X_MICROARCH.md: host link, PCIe Gen5 x16 (spec)
VALU_LANE_OPS = 256 * 4 * 32 * 2.4e9  # 256 CUs x 4 SIMD-32 x 2.4 GHz lane-instructions/s
# The exact cell is add, add, 1/2 max3; tools/ubench2: v_add_f32 issues in 2 cycles per wave, v_max3_f32 in 4 -> 4 issue
# cycles per cell and wave = 16 cells per cycle and SIMD.  The ceiling of any kernel that evaluates its cells this way:
CELLS_PER_CYCLE = 256 * 4 * 16        # x the shader clock DELIVERED under the kernel's load (torbi_hip_scan_stats [120], [121])


def valu_ceiling(clock_hz):
    """Cells per second the vector ALUs can evaluate at `clock_hz` (None: the data sheet's 2.4 GHz)."""
    return CELLS_PER_CYCLE * (clock_hz or 2.4e9)
METRIC = 'timesteps decoded/sec, 1440 states batch=512'
KERNELS = {'resident': 'resident::resident_forward_kernel', 'cluster': 'resident::resident_forward_kernel',
           'dense': 'dense::step_dense_kernel', 'generic': 'step_rows', 'rows': 'rowscan::step_rows_sorted_kernel'}
ROUTES = {0: 'generic', 1: 'dense', 3: 'resident', 4: 'rows', 5: 'cluster', 6: 'held', 7: 'small', 8: 'band'}


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=16)
    ap.add_argument('--warmup', type=int, default=8)
    ap.add_argument('--workload', choices=['c3', 'c4'], default='c3',
                    help='c3 = BASELINE configs[2] (headline, weak scaling); c4 = configs[3]: ragged many-file job, '
                         'lengths 100..900, batches of 512 padded to the batch maximum, sharded over the ranks '
                         '(strong scaling; --steps limits the number of batches, 0 = all of them)')
    ap.add_argument('--files', type=int, default=40000, help='sequences of the c4 job')
    ap.add_argument('--end-to-end', type=int, default=0,
                    help='c4 only: additionally decode this many sequences from torch.save()d files to files '
                         '(torch.load + H2D + decode + D2H + torch.save), reported under "end_to_end"')
    ap.add_argument('--batch', type=int, default=512)
    ap.add_argument('--frames', type=int, default=500)
    ap.add_argument('--states', type=int, default=1440)
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-secondary', action='store_true', help='skip the secondary records (other configs / inputs)')
    ap.add_argument('--no-single-call', action='store_true', help='skip the single_call record (profiling runs: nothing but '
                                                                  'the launch groups in the trace)')
    ap.add_argument('--reuse-preparation', action='store_true',
                    help='let consecutive decodes share the per-transition preparation (sorted rows / packed panels) '
                         'as a serving loop would; off by default: every timed launch group does all of its work')
    ap.add_argument('--forward', choices=['auto', 'dense', 'resident', 'cluster'], default='auto',
                    help='forward-recurrence path (include/torbi_hip.h); every path gives identical indices')
    ap.add_argument('--pipeline', type=int, default=2, help='HIP streams the launch groups alternate between')
    ap.add_argument('--groups', choices=['balanced', 'full'], default='full',
                    help='how --steps batches are cut into launch groups: balanced = equally full groups (20 -> 7 + 7 + 6), '
                         'full = as many full groups as possible, the rest as one smaller group (20 -> 8 + 8 + 4; a group below '
                         'half the chip runs in the cluster form)')
    ap.add_argument('--group', type=int, default=8,
                    help='batches decoded per launch group (1 = every batch on its own: per-timestep launches)')
    ap.add_argument('--transition', choices=['dense', 'banded', 'uniform'], default='dense',
                    help="dense = headline workload; banded = the reference's pitch transition "
                         '(torbi/evaluate/core.py:24-33); uniform = the reference default (transition=None)')
    ap.add_argument('--half-width', type=float, default=87.2,
                    help='band half width in states for --transition banded (penn: 87.2)')
    return ap.parse_args(argv)


def self_launch(args):
    """--gpus N > 1 without a launcher: start the N ranks ourselves (one process per GPU, rendezvous on
    127.0.0.1) as a CHILD process -- nothing in this process has touched a GPU -- and leave with its code."""
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', f'--nproc-per-node={args.gpus}',
           '--master-addr', '127.0.0.1', '--master-port', str(port), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.call(cmd)


def algorithmic_bytes_per_timestep(S):
    """4S observation read + 4S int32 backpointer write + 8 (backtrace)."""
    return 8 * S + 8


def profiled_traffic(kernel_name, batches):
    """HBM-side bytes per launch of the dominant kernel from the newest committed rocprofv3 PMC summary that was taken ON
    THE KERNEL THAT IS RUNNING (profiles/rNN_pmc.json: separate --pmc FETCH_SIZE / WRITE_SIZE passes of this same command;
    `kernel_name` = torbi_hip_last_forward_kernel(), the template instance as rocprofv3 spells it).  FETCH_SIZE is
    doubled as MI355X_MICROARCH.md prescribes for wide coalesced reads on gfx950; both counters are in KiB and include
    Infinity-Cache hits.  The summary was taken with `_meta.batches_per_forward_launch` batches per launch; the figure is
    scaled to the `batches` this run puts into one.  Returns (bytes or None, provenance): a summary that names another
    instance of the kernel (other template arguments, i.e. other code) is NOT used -- the line then says so instead of
    reporting stale traffic."""
    folder = os.path.join(ROOT, 'profiles')
    squeeze = lambda name: name.replace(' ', '')
    try:
        names = sorted(f for f in os.listdir(folder) if f.endswith('_pmc.json'))
    except OSError:
        return None, {'file': None, 'reason': 'no profiles/ folder'}
    for name in reversed(names):
        try:
            pmc = json.load(open(os.path.join(folder, name)))
        except (OSError, ValueError):
            continue
        meta = pmc.get('_meta', {})
        taken_with = float(meta.get('batches_per_forward_launch', batches))
        for kernel, counters in pmc.items():
            if squeeze(kernel) == squeeze(kernel_name) and 'FETCH_SIZE' in counters and 'WRITE_SIZE' in counters:
                nbytes = (2.0 * counters['FETCH_SIZE']['mean_per_dispatch']
                          + counters['WRITE_SIZE']['mean_per_dispatch']) * 1024.0 * batches / taken_with
                mean = lambda c: counters[c]['mean_per_dispatch'] if c in counters else None
                return nbytes, {'file': f'profiles/{name}', 'kernel': kernel, 'taken_at_commit': meta.get('git'),
                                'command': meta.get('command'),
                                'batches_per_launch_when_taken': int(taken_with),
                                'dispatches': counters['FETCH_SIZE'].get('dispatches'),
                                # the same passes' shader counters, per dispatch of `taken_with` batches (None: not collected)
                                'counters': {c: mean(c) for c in ('SQ_INSTS_VALU', 'SQ_LDS_IDX_ACTIVE', 'SQ_LDS_BANK_CONFLICT',
                                                                  'SQ_BUSY_CYCLES', 'SQ_WAVE_CYCLES')},
                                'executed_cells_per_launch_when_taken': meta.get('executed_cells_per_launch')}
    return None, {'file': None, 'running_kernel': kernel_name,
                  'reason': 'no committed PMC summary was taken on this kernel instance'}


def cpu_baseline(obs_dev, trans_dev, init_dev, gpu_indices, budget_s=15.0):
    """Time the CPU path on this host's cores on a bounded sample of the SAME workload and check the GPU's
    indices against it.  Uses the reference's own operator (oracle/_ref, kind "reference") when that build is
    present, else the reference-shaped C port (oracle mode 0, kind "port").  The thread count is calibrated on a
    16-frame prefix (the reference's at::parallel_for over states gets SLOWER with hundreds of threads), the
    sample is sized to about `budget_s` seconds, and the one-thread rate is measured beside it."""
    import numpy as np
    import oracle
    host = os.cpu_count() or 1
    T, S = obs_dev.shape[1], obs_dev.shape[2]
    trans = trans_dev.cpu().numpy()
    init = init_dev.cpu().numpy()
    try:
        use_ref = oracle.ref_available()
        if use_ref:
            oracle.ref_decode(np.zeros((1, 2, 2), np.float32), [2], np.zeros((2, 2), np.float32),
                              np.zeros(2, np.float32))
    except Exception:
        use_ref = False

    def run(obs, frames, threads):
        t0 = time.perf_counter()
        if use_ref:
            idx = oracle.ref_decode(obs, frames, trans, init, num_threads=threads).numpy()
        else:
            idx = oracle.decode(obs, frames, trans, init, num_threads=threads, mode=0)
        return time.perf_counter() - t0, idx

    item0 = obs_dev[:1].cpu().numpy()
    cal = np.ascontiguousarray(item0[:, :16])
    best_threads, best_dt, one_dt = 1, None, None
    for threads in (1, 2, 4, 8, 16, 32, 64, 128):
        if threads > host:
            break
        dt = min(run(cal, np.array([16], np.int32), threads)[0] for _ in range(2))     # best of two: the pool warms up
        if threads == 1:
            one_dt = dt
        if best_dt is None or dt < best_dt:
            best_threads, best_dt = threads, dt
        elif dt > 2 * best_dt:
            break
    # one thread: a longer prefix of the same item (15 -> 63 recurrence steps), a few seconds at most
    n1 = 64 if one_dt * 4 < 8.0 else 16
    dt1, _ = run(np.ascontiguousarray(item0[:, :n1]), np.array([n1], np.int32), 1)
    # the sample: items in chunks until the budget is spent (the batch loop of the reference is serial, viterbi.cpp:65,
    # so chunking does not change its rate; a mis-calibrated estimate cannot run away with the bench)
    per_item = best_dt / 15.0 * (T - 1)            # 15 recurrence steps in the prefix
    chunk = int(max(1, min(32, budget_s / 8 / max(per_item, 1e-3))))
    items, dt, match = 0, 0.0, True
    while items < obs_dev.shape[0] and dt < budget_s:
        n = min(chunk, obs_dev.shape[0] - items)
        obs = obs_dev[items:items + n].cpu().numpy()
        step_dt, idx = run(obs, np.full((n,), T, np.int32), best_threads)
        match = match and bool(np.array_equal(idx, gpu_indices[items:items + n].cpu().numpy()))
        items += n
        dt += step_dt
    return {
        'value': items * T / dt, 'unit': 'timesteps/s', 'cores': best_threads,
        'kind': 'reference' if use_ref else 'port',
        'sample': f'first {items} of {obs_dev.shape[0]} items x {T} frames x {S} states, '
                  f'{best_threads} threads (best of a 16-frame calibration; host has {host} '
                  f'logical CPUs), {dt:.1f} s',
        'gpu_matches_cpu': match,
        'one_thread': {'value': n1 / dt1, 'unit': 'timesteps/s', 'cores': 1,
                       'sample': f'first {n1} frames of item 0, {dt1:.1f} s'},
    }


class Bench:
    """Everything that needs torch; constructed after the launch decisions."""

    def __init__(self, args):
        import numpy as np
        import torch
        import torch.distributed as dist
        import torbi_amd
        from torbi_amd import distributed, synth, viterbi
        self.np, self.torch, self.dist = np, torch, dist
        self.torbi_amd, self.distributed, self.synth, self.viterbi = torbi_amd, distributed, synth, viterbi
        self.args = args
        self.dry = torch.cuda.device_count() == 0
        backend = 'gloo' if self.dry else None
        self.rank, self.size, self.local = distributed.init_from_env(backend)
        assert self.size == args.gpus, f'--gpus {args.gpus} but WORLD_SIZE={self.size}'
        self.collective = dist.is_available() and dist.is_initialized()
        self.dev = None if self.dry else torch.device('cuda', torch.cuda.current_device())

    # ---- helpers --------------------------------------------------------------------------------------
    def fence(self, pipe=None):
        torch = self.torch
        if pipe is not None:
            pipe.synchronize()
        if not self.dry:
            torch.cuda.synchronize()
        if self.collective:
            self.dist.barrier()
        if not self.dry:
            torch.cuda.synchronize()

    def max_over_ranks(self, seconds):
        if not self.collective:
            return seconds
        t = self.torch.tensor([seconds], dtype=self.torch.float64, device=self.dev or 'cpu')
        self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX)
        return float(t.item())

    def sum_over_ranks(self, value):
        if not self.collective:
            return value
        t = self.torch.tensor([float(value)], dtype=self.torch.float64, device=self.dev or 'cpu')
        self.dist.all_reduce(t, op=self.dist.ReduceOp.SUM)
        return float(t.item())

    def rank_report(self, local_rate, alone, indices, B):
        """What makes an N > 1 line verifiable from its own output (rates per GPU count, gather time shown
        separately): the world as the collective saw it, every rank's own rate inside the timed region, the all-gather
        of one batch's indices timed on its own, and rank 0's rate over the same steps with NO collective (the driver's
        N = 1 run should agree with it)."""
        torch, dist = self.torch, self.dist
        size = self.size
        ones = int(round(self.sum_over_ranks(1.0)))
        report = {'ranks_seen': {'world_size': dist.get_world_size(), 'all_reduce_of_ones': ones,
                                 'backend': dist.get_backend()}}
        mine = torch.tensor([local_rate if local_rate is not None else float('nan')], dtype=torch.float64,
                            device=self.dev or 'cpu')
        every = [torch.zeros_like(mine) for _ in range(size)]
        dist.all_gather(every, mine)
        report['per_rank_value'] = [None if math.isnan(float(v)) else float(v) for v in every]
        # the collective alone: K all-gathers of one batch's (B, T) int32 indices, max over ranks
        rounds = 20
        self.fence()
        t0 = time.perf_counter()
        for _ in range(rounds):
            self.distributed.gather_indices(indices, B * size, force=True)
        if not self.dry:
            torch.cuda.synchronize()
        report['gather_ms_per_batch'] = self.max_over_ranks(time.perf_counter() - t0) / rounds * 1e3
        report['gather_bytes_per_batch'] = int(indices.numel()) * 4 * size
        if alone is not None:
            steps = min(self.args.steps, 8)
            alone(steps)                         # (warm: the pipeline's streams without the after-hook)
            self.fence()
            t0 = time.perf_counter()
            alone(steps)
            if not self.dry:
                torch.cuda.synchronize()
            rate = float(indices.shape[0]) * indices.shape[1] * steps / (time.perf_counter() - t0)
            alone_rates = [torch.zeros_like(mine) for _ in range(size)]
            dist.all_gather(alone_rates, torch.tensor([rate], dtype=torch.float64, device=self.dev or 'cpu'))
            report['n1_reference_value'] = float(alone_rates[0])
            report['per_rank_value_without_collective'] = [float(v) for v in alone_rates]
            report['n1_reference_note'] = (f'rank 0 over {steps} steps with no collective while every other rank does the '
                                           'same on its own GPU: what `bench.py --gpus 1` measures, up to launch-group shape')
            self.fence()
        else:
            report['n1_reference_value'] = None
        return report

    def model(self, S, transition='dense', half_width=87.2):
        v, synth = self.viterbi, self.synth
        trans = v.fill_synthetic((S, S), synth.STREAM_TRANSITION, seed=0, device=self.dev)
        if transition == 'banded':
            trans = self.torch.from_numpy(synth.banded_transition(S, half_width)).to(self.dev)
        init = v.fill_synthetic((S,), synth.STREAM_INITIAL, seed=0, device=self.dev)
        return trans, init

    def balanced_groups(self, steps, group):
        """Group sizes for `steps` consecutive batches: as few launch groups as `group` allows, equally full
        (20 steps, group 8 -> 7 + 7 + 6 rather than 8 + 8 + 4: a half-empty last group costs a full one's time)."""
        if group <= 1:
            return [1] * steps
        if getattr(self.args, 'groups', 'balanced') == 'full':
            return [group] * (steps // group) + ([steps % group] if steps % group else [])
        n = max(1, math.ceil(steps / group))
        base, extra = divmod(steps, n)
        return [base + (1 if k < extra else 0) for k in range(n)]

    def timed_decodes(self, decode_one, count, warmup=1):
        """Median-free plain timing of `count` serial decodes (secondary records): seconds per decode."""
        torch = self.torch
        for _ in range(warmup):
            decode_one()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(count):
            out = decode_one()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / count, out

    # ---- headline: BASELINE configs[2] ----------------------------------------------------------------
    def run_c3(self):
        args, torch, v, synth = self.args, self.torch, self.viterbi, self.synth
        B, T, S = args.batch, args.frames, args.states
        rank, size = self.rank, self.size
        if self.dry:
            self.fence()
            line = {'metric': METRIC, 'value': None, 'unit': 'timesteps/s', 'n_gpus': size, 'steps': args.steps,
                    'warmup': args.warmup, 'ms_per_step': None, 'higher_is_better': True, 'scaling': 'weak',
                    'vs_baseline': None, 'dtype': 'f32', 'data': 'synthetic', 'dry_run': True,
                    'config': {'workload': f'{S} states, {T} frames, batch={B} per GPU (no HIP device: launch, '
                                           f'rendezvous and planning only)',
                               'launch_groups': self.balanced_groups(args.steps, args.group)}}
            if self.collective:
                line['multi_gpu'] = self.rank_report(None, None, torch.zeros((B, T), dtype=torch.int32), B)
            return line
        dev = self.dev
        trans, init = self.model(S, args.transition, args.half_width)
        frames = torch.full((B,), T, dtype=torch.int32, device=dev)
        uniform_c = float(torch.tensor(math.log(1.0 / S), dtype=torch.float32))
        group = 1 if args.transition == 'uniform' else max(1, args.group)
        # distinct observations for the batches of a launch group (rank-specific streams; shared model)
        obs = [v.fill_synthetic((B, T, S), synth.STREAM_OBSERVATION, seed=rank * 64 + k, device=dev)
               for k in range(group)]
        path = None if args.forward == 'auto' else args.forward
        pipe = self.torbi_amd.DecodePipeline(dev, depth=max(1, args.pipeline), reuse_preparation=args.reuse_preparation,
                                             group=group, path=path)

        pipe.reserve(B, T, S)          # workspace allocation is not part of a decode

        def gather(idx):
            return self.distributed.gather_indices(idx, B * size, force=True) if self.collective else idx

        def run_steps(count):
            last, k = None, 0
            for n in self.balanced_groups(count, group):
                for j in range(n):
                    if args.transition == 'uniform':
                        last = gather(self.torbi_amd.decode_uniform(obs[0], frames, uniform_c, init))
                    else:
                        last = pipe.decode(obs[j % group], frames, trans, init, after=gather)
                    k += 1
                pipe.flush()
            return last

        run_steps(args.warmup)
        self.fence(pipe)
        t0 = time.perf_counter()
        indices = run_steps(args.steps)
        self.fence(pipe)
        local_elapsed = time.perf_counter() - t0
        elapsed = self.max_over_ranks(local_elapsed)
        value = float(B) * T * args.steps * size / elapsed
        last_obs = obs[(self.balanced_groups(args.steps, group)[-1] - 1) % group]

        result = {
            'metric': METRIC, 'value': value, 'unit': 'timesteps/s', 'n_gpus': size, 'steps': args.steps,
            'warmup': args.warmup, 'ms_per_step': elapsed / args.steps * 1e3, 'higher_is_better': True,
            'scaling': 'weak', 'vs_baseline': None, 'dtype': 'f32', 'data': 'synthetic',
        }
        if self.collective:
            def alone(count):                  # the same steps WITHOUT the collective: what this rank does on its own
                pipe.synchronize()
                for n in self.balanced_groups(count, group):
                    for j in range(n):
                        if args.transition == 'uniform':
                            self.torbi_amd.decode_uniform(obs[0], frames, uniform_c, init)
                        else:
                            pipe.decode(obs[j % group], frames, trans, init)
                    pipe.flush()
                pipe.synchronize()
            result['multi_gpu'] = self.rank_report(float(B) * T * args.steps / local_elapsed, alone, indices, B)
        if args.transition == 'uniform':
            per = elapsed / args.steps
            nbytes = float(B) * T * (4 * S + 4)
            result['config'] = {'workload': f'{S} states, {T} frames, batch={B} per GPU, fp32, UNIFORM transition '
                                            f'(the reference default, transition=None; secondary workload)'}
            result['roofline'] = {'bound': 'hbm', 'achieved': nbytes / per / 1e9, 'peak': HBM_PEAK_GBS, 'unit': 'GB/s',
                                  'frac': nbytes / per / 1e9 / HBM_PEAK_GBS, 'traffic': None,
                                  'kernel': 'uniform::uniform_rows_kernel (whole decode, host-timed incl. launch)',
                                  'algorithmic_bytes_per_launch': nbytes}
            return result

        # dominant kernel, measured live with hipEvents on the launch stream (include/torbi_hip.h: phase_ms) on a
        # full launch group, averaged over three profiled groups
        sizes = self.balanced_groups(args.steps, group)
        g = max(sizes)
        spaces = [torch.empty(v.workspace_bytes(B, T, S), dtype=torch.uint8, device=dev) for _ in range(g)]
        prof, fwd_ms, bt_ms, prep_ms = [], 0.0, 0.0, 0.0
        for _ in range(3):
            v.decode_batches([obs[k % group] for k in range(g)], [frames] * g, trans, init, workspaces=spaces,
                             path=path, _profile=prof)
            fwd_ms += prof[0] / 3
            bt_ms += prof[1] / 3
            prep_ms += prof[4] / 3
        route = ROUTES[int(prof[3])]
        launches = max(int(prof[2]), 1)
        covered = max(int(prof[5]), 1)                     # batches one forward launch (chain) covers
        kernel_s = (fwd_ms - prep_ms) * 1e-3 / launches
        timesteps_per_launch = float(covered) * B * (T if route in ('resident', 'cluster', 'band') else 1)
        bytes_per_launch = timesteps_per_launch * algorithmic_bytes_per_timestep(S)
        achieved = bytes_per_launch / kernel_s / 1e9
        cells_per_launch = timesteps_per_launch * S * S
        running = v.last_forward_kernel()          # the template instance the profiled groups launched
        traffic, provenance = profiled_traffic(running, covered)
        # executed work, live: the kernel counts the 16-entry list blocks its sampled wave passes walk (torbi_hip_scan_stats)
        executed = None
        if route in ('resident', 'cluster'):
            stats = v.scan_stats(spaces[0], B, T, S, path='resident').cpu().to(torch.int64)
            blocks = float(stats[:64].sum()) / max(1.0, float(stats[64:120].sum()))
            measured = v.delivered_clock_hz(stats)
            ni = 16 if S <= 2048 else 8
            rows_per_pass = 64 // (ni // 4)
            passes = covered * math.ceil(B / ni) * math.ceil(S / rows_per_pass) * (T - 1)
            cells = passes * blocks * 16.0 * rows_per_pass * ni
            clock = measured or 2.4e9
            # per-cell costs from the committed PMC summary of THIS kernel instance (profiled_traffic matched it by name):
            # vector instructions per examined cell and lane, and the LDS's bank-conflict factor; None when no summary matches
            pmc = (provenance or {}).get('counters') or {}
            scale = covered / float((provenance or {}).get('batches_per_launch_when_taken') or covered)
            pmc_cells = (provenance or {}).get('executed_cells_per_launch_when_taken')
            pmc_cells = pmc_cells * scale if pmc_cells else cells          # (the same seeded inputs: the same cells)
            instr = pmc['SQ_INSTS_VALU'] * scale * 64.0 / pmc_cells if pmc.get('SQ_INSTS_VALU') and pmc_cells else None
            conflict = (1.0 + pmc['SQ_LDS_BANK_CONFLICT'] / (pmc['SQ_LDS_IDX_ACTIVE'] - pmc['SQ_LDS_BANK_CONFLICT'])
                        if pmc.get('SQ_LDS_IDX_ACTIVE') and pmc.get('SQ_LDS_BANK_CONFLICT') is not None
                        and pmc['SQ_LDS_IDX_ACTIVE'] > pmc['SQ_LDS_BANK_CONFLICT'] else None)
            executed = {
                'clock_hz_measured': measured,
                'list_blocks_per_wave_pass': blocks, 'row_blocks': math.ceil(S / 16),
                'cells_per_launch': cells, 'cells_per_s': cells / kernel_s,
                'fraction_of_all_cells': cells / cells_per_launch if cells_per_launch else None,
                # issue model of the scan: per entry pair and lane 4 quad broadcasts + 8 adds + 4 max3 = 16 instructions
                # for 8 cells (tools/ubench2: v_max3_f32 / DPP moves ~4, v_add_f32 2 cycles per wave instruction: 2.5 on
                # average); + bound test, list loads, epilogue
                'valu_instr_per_cell': instr,
                'lds_conflict_factor': conflict,
                'per_cell_costs': ('SQ_INSTS_VALU x 64 lanes / examined cells and SQ_LDS_IDX_ACTIVE / (SQ_LDS_IDX_ACTIVE - '
                                   'SQ_LDS_BANK_CONFLICT) of the committed PMC summary named in roofline.traffic_source (taken on '
                                   'this kernel instance); list_blocks_per_wave_pass and the clock are measured in this run')
                if instr else 'None: no committed PMC summary was taken on the running kernel instance',
                'valu_busy_frac': cells * instr * 2.5 / 64.0 / (256 * 4 * clock * kernel_s) if instr else None,
                # one ds_read_b128 per 4 cells and lane = 4 LDS cycles per wave instruction x the conflict factor
                'lds_busy_frac': cells / 256.0 * 4.0 * conflict / (256 * clock * kernel_s) if conflict else None,
                'statistics_gave_up': int(stats[127]),
                'note': 'list blocks per wave pass live from torbi_hip_scan_stats (every 16th timestep sampled) x 16 '
                        'entries x 256 (row, item) pairs; the busy fractions price those cells with the per-cell costs '
                        'stated here at the measured clock (2.5 issue cycles per vector instruction on average) -- both pipes '
                        'are more than half busy and do not overlap fully: that, not HBM, is what binds this kernel'}
        del spaces
        result['config'] = {
            'workload': (f'{S} states, {T} frames, batch={B} per GPU, fp32, dense transition'
                         f'{" (BASELINE configs[2])" if (B, T, S) == (512, 500, 1440) else ""}; decode = forward + '
                         f'argmax + backtrace, inputs resident in HBM') if args.transition == 'dense' else
                        (f'{S} states, {T} frames, batch={B} per GPU, fp32, BANDED transition (half width '
                         f'{args.half_width}, -inf outside; secondary workload)'),
            'parallelism': f'batch-sharded x{size}' if size > 1 else 'single GPU',
            'launch_groups': sizes, 'streams': args.pipeline, 'forward_path': route,
            'throughput_mode': f'{sum(sizes)} batches decoded as {len(sizes)} launch groups over {args.pipeline} '
                               f'streams; ONE call on ONE batch is "single_call"',
            'transition_preparation': 'reused across launch groups' if args.reuse_preparation
            else 'rebuilt by every launch group'}
        result['roofline'] = {
            'bound': 'valu_issue+lds' if route in ('resident', 'cluster') else 'valu_issue',
            'bound_note': 'achieved / peak / frac are the HBM figures BASELINE.json asks for (algorithmic bytes per launch '
                          'over the 8 TB/s peak); what binds the kernel is under "executed"',
            'achieved': achieved, 'peak': HBM_PEAK_GBS, 'unit': 'GB/s',
            'frac': achieved / HBM_PEAK_GBS, 'traffic': traffic if (B, T, S) == (512, 500, 1440) else None,
            'traffic_source': provenance,
            'traffic_note': '2 * FETCH_SIZE + WRITE_SIZE of that kernel instance, Infinity-Cache hits included, scaled to '
                            f'{covered} batches per launch; above the algorithmic bytes: the sorted transition lists and seed '
                            'rows that miss the 4 MB L2s (served by the Infinity Cache)',
            'kernel': running + (f' (ONE launch = the whole forward pass of {covered} batches)'
                                 if route in ('resident', 'cluster', 'band') else ' (one launch = one timestep of one batch)'),
            'launch_us': kernel_s * 1e6, 'launches_per_group': launches, 'batches_per_launch': covered,
            'algorithmic_bytes_per_launch': bytes_per_launch,
            'executed': executed,
            'note': 'the (max,+) recurrence holds S/4 = 360 cells per algorithmic byte: kernels that evaluate every cell '
                    'are VALU-bound; the pruned recurrence is bound by LDS gathers + VALU issue (DESIGN.md 4)'}
        result['valu'] = {
            'dense_equivalent_cells_per_s': cells_per_launch / kernel_s, 'lane_instr_peak_per_s': VALU_LANE_OPS,
            'clock_hz_measured': (executed or {}).get('clock_hz_measured'),
            'ceiling_cells_per_s_at_measured_clock': valu_ceiling((executed or {}).get('clock_hz_measured')),
            'ceiling_note': 'add, add, 1/2 max3 per cell = 4 issue cycles per wave (v_add_f32 2, v_max3_f32 4: tools/ubench2) = 16 '
                            'cells per cycle and SIMD x 1024 SIMDs x the clock delivered under the forward kernel\'s own load '
                            '(39.3 Tcell/s at the data sheet\'s 2.4 GHz)',
            'note': 'dense-equivalent = every (prev, next) cell of the launch, pruned or not, over the kernel time: a '
                    'speed-up figure against kernels that evaluate every cell, NOT a utilisation (roofline.executed has that)'}
        result['phases_ms'] = {'group_of': g, 'forward_incl_preparation': fwd_ms, 'preparation': prep_ms,
                               'argmax_backtrace': bt_ms}
        result['hbm_roofline_frac_whole_job'] = value / size * algorithmic_bytes_per_timestep(S) / (HBM_PEAK_GBS * 1e9)
        if rank == 0 and size == 1 and args.transition == 'dense' and not args.no_single_call:
            result['single_call'] = self.single_call(obs[0], frames, trans, init)
        if rank == 0 and size == 1 and not args.no_secondary and args.transition == 'dense':
            result['secondary'] = self.secondary(obs[0], frames, trans, init)
            if 'single_call' in result:
                result['secondary']['serial'] = dict(result['single_call'], note='= single_call (kept under its old name)')
        if rank == 0 and size == 1 and not args.no_cpu_baseline:
            result['cpu_baseline'] = cpu_baseline(last_obs, trans, init, indices)
        # every BASELINE config and the structured-transition routes, compact, INSIDE the part of the line the driver keeps
        sec_, one = result.get('secondary') or {}, result.get('single_call')

        def compact(rec, roof_timesteps_per_s, **more):
            if not rec or rec.get('value') is None:
                return None
            found = {'value': rec['value'], 'ms_per_decode': rec.get('ms_per_decode'),
                     'frac': rec['value'] / roof_timesteps_per_s, 'kernel': rec.get('kernel'),
                     'forward_path': rec.get('forward_path')}
            found.update(more)
            return found
        roof = lambda S_: HBM_PEAK_GBS * 1e9 / algorithmic_bytes_per_timestep(S_)
        band_exec = lambda rec: {'frac_of_alu_ceiling_at_measured_clock':
                                 ((rec or {}).get('executed') or {}).get('forward_frac_of_ceiling_at_measured_clock')}
        configs = {
            'c2': compact(sec_.get('c2'), roof(S)),
            'c3_launch_group': {'value': value / size, 'ms_per_decode': elapsed / args.steps * 1e3,
                                'frac': result['hbm_roofline_frac_whole_job'], 'kernel': running, 'forward_path': route},
            'c3_single_call': compact(one, roof(S)),
            'c3_every_cell': compact(sec_.get('every_cell'), roof(S),
                                     frac_of_alu_ceiling_at_measured_clock=(sec_.get('every_cell') or {}).get('frac_of_ceiling_at_measured_clock')),
            'c3_peaked_dense': compact(sec_.get('peaked_dense_transition'), roof(S)),
            'c4_decode_only': compact(sec_.get('c4_40000_files'), roof(S)),
            'c5': compact(sec_.get('c5'), roof(4096)),
            'band_single': compact(sec_.get('peaked_banded'), roof(S), **band_exec(sec_.get('peaked_banded'))),
            'band_group': compact(sec_.get('peaked_banded_launch_group'), roof(S),
                                  **band_exec(sec_.get('peaked_banded_launch_group'))),
            'band_group_evaluation_matrix': compact(sec_.get('evaluation_matrix_launch_group'), roof(S),
                                                    **band_exec(sec_.get('evaluation_matrix_launch_group'))),
            'band_single_evaluation_matrix': compact(sec_.get('evaluation_matrix'), roof(S),
                                                     **band_exec(sec_.get('evaluation_matrix'))),
            'uniform': compact(sec_.get('uniform'), HBM_PEAK_GBS * 1e9 / (4 * S + 4)),
        }
        result['roofline']['configs'] = {k: c for k, c in configs.items() if c}
        result['roofline']['configs_note'] = ('value in timesteps/s; frac = value x algorithmic bytes per timestep (8 S + 8; uniform: '
                                              '4 S + 4) / 8 TB/s; c2 / c3_* / c5 = BASELINE configs[1], [2], [4] (c3_launch_group = the '
                                              'headline value, c3_single_call = ONE decode of ONE batch, c3_every_cell = the dense '
                                              'kernel forced, c3_peaked_dense = posteriorgram-like rows); band_* = the reference\'s pitch '
                                              'transition (torbi/evaluate/core.py:24-33) on peaked rows, log(p) (-inf outside the band) '
                                              'and, *_evaluation_matrix, log(p + tiny) as torbi.evaluate really decodes; the long notes are under '
                                              '"secondary"')
        return result

    # ---- the literal BASELINE config: ONE decode call on ONE batch -------------------------------------
    def single_call(self, obs, frames, trans, init):
        """What the reference times (torbi/core.py:200-206): ONE `decode` of ONE batch, nothing else in flight.  The
        workspace is the caller's and allocated once; the per-transition preparation is rebuilt by every
        call unless --reuse-preparation.  Median of 5 after 3 warm-up calls."""
        torch, v = self.torch, self.viterbi
        B, T, S = obs.shape
        ws = torch.empty(v.workspace_bytes(B, T, S), dtype=torch.uint8, device=self.dev)
        reuse = bool(self.args.reuse_preparation)
        prof = []
        for _ in range(3):
            self.torbi_amd.decode(obs, frames, trans, init, workspace=ws, reuse_preparation=reuse, _profile=prof)
        kernel = v.last_forward_kernel()
        times = []
        for _ in range(5):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            self.torbi_amd.decode(obs, frames, trans, init, workspace=ws, reuse_preparation=reuse)
            torch.cuda.synchronize()
            times.append(time.perf_counter() - t0)
        sec = sorted(times)[len(times) // 2]
        rate = B * T / sec
        kept = []
        for _ in range(6):          # ... and as a serving loop with ONE matrix makes the call (DecodePipeline does)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            self.torbi_amd.decode(obs, frames, trans, init, workspace=ws, reuse_preparation=True)
            torch.cuda.synchronize()
            kept.append(time.perf_counter() - t0)
        sec_kept = sorted(kept[1:])[2]
        plain = []
        for _ in range(6):          # ... and exactly as the reference's caller writes it: no workspace argument at all
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            self.torbi_amd.decode(obs, frames, trans, init)
            torch.cuda.synchronize()
            plain.append(time.perf_counter() - t0)
        sec_plain = sorted(plain[1:])[2]
        return {'value': rate, 'unit': 'timesteps/s', 'ms_per_decode': sec * 1e3,
                'without_workspace_argument': {'value': B * T / sec_plain, 'ms_per_decode': sec_plain * 1e3,
                                               'note': 'decode(observation, batch_frames, transition, initial) as '
                                                       'torbi/core.py:200-206 calls it: scratch from the caching allocator '
                                                       'per call, the preparation kept with the transition tensor '
                                                       '(torbi_hip_viterbi_decode_batches_prepared); first call excluded'},
                'with_reused_preparation': {'value': B * T / sec_kept, 'ms_per_decode': sec_kept * 1e3,
                                            'note': 'decode(..., workspace=ws, reuse_preparation=True): sorted rows / '
                                                    'transposed matrix taken from the workspace of the previous call with '
                                                    'the same matrix (TORBI_HIP_REUSE_TRANSITION)'},
                'roofline_frac': rate * algorithmic_bytes_per_timestep(S) / (HBM_PEAK_GBS * 1e9),
                'forward_path': ROUTES[int(prof[3])], 'kernel': kernel,
                'phases_ms': {'forward_incl_preparation': prof[0], 'preparation': prof[4], 'argmax_backtrace': prof[1]},
                'us_per_timestep_forward': (prof[0] - prof[4]) / max(T - 1, 1) * 1e3,
                'note': f'ONE torbi_amd.decode() of ONE {B} x {T} x {S} batch resident in HBM, host-timed around a '
                        'synchronised call (median of 5), workspace preallocated, transition preparation '
                        + ('reused' if reuse else 'rebuilt every call')}

    # ---- secondary records: the same run substantiates DESIGN.md's table -------------------------------
    def secondary(self, obs, frames, trans, init):
        torch, v, synth, np = self.torch, self.viterbi, self.synth, self.np
        args, dev = self.args, self.dev
        B, T, S = obs.shape
        out = {}

        def record(name, seconds, timesteps, S_, note, extra=None):
            rate = timesteps / seconds
            out[name] = {'value': rate, 'unit': 'timesteps/s', 'ms_per_decode': seconds * 1e3,
                         'roofline_frac': rate * algorithmic_bytes_per_timestep(S_) / (HBM_PEAK_GBS * 1e9),
                         'note': note}
            if extra:
                out[name].update(extra)

        # BASELINE configs[3] on this one GPU: 40 000 ragged sequences (79 batches, ten launch groups), decode only
        out['c4_40000_files'] = self.c4_decode_only(40000, steps=0, quiet=True)
        ws = torch.empty(v.workspace_bytes(B, T, S), dtype=torch.uint8, device=dev)
        prof = []
        for _ in range(3):
            self.torbi_amd.decode(obs, frames, trans, init, workspace=ws, _profile=prof)
        # the call a drop-in user makes (reference torbi/core.py:110-208): device-resident log-probabilities through
        # from_probabilities -- epsilon round trip (in place, like upstream), workspace allocation and decode included
        samples = []
        for k in range(4):
            x = obs.clone()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            self.torbi_amd.from_probabilities(x, frames, trans, init, log_probs=True, gpu=dev.index or 0)
            torch.cuda.synchronize()
            samples.append(time.perf_counter() - t0)
            del x
        sec = sorted(samples[1:])[1]
        record('api_from_probabilities', sec, B * T, S,
               'torbi_amd.from_probabilities(observation[B,T,S] on the device, batch_frames, transition, initial, '
               'log_probs=True, gpu=0): epsilon clamp pass over the batch + workspace allocation (caching allocator) + '
               'decode, host-timed around a synchronised call, median of 3 after one warm-up call',
               {'first_call_ms': samples[0] * 1e3})
        # the reference's call with EVERY default: probabilities in, no transition, no initial (uniform transition)
        probs = torch.softmax(obs, dim=-1)
        sec, _ = self.timed_decodes(lambda: self.torbi_amd.from_probabilities(probs, frames, gpu=0), 3)
        tiny = torch.finfo(torch.float32).tiny
        c0 = float(torch.tensor(math.log(1.0 / S), dtype=torch.float32))
        i0 = torch.full((S,), math.log(1.0 / S + tiny), dtype=torch.float32, device=dev)
        sec_steps, _ = self.timed_decodes(lambda: self.torbi_amd.decode_uniform(v.log_epsilon_clamp(probs), frames, c0, i0), 3)
        out['api_from_probabilities_all_defaults'] = {
            'value': B * T / sec, 'unit': 'timesteps/s', 'ms_per_decode': sec * 1e3,
            'roofline_frac': B * T * (4 * S + 4) / sec / (HBM_PEAK_GBS * 1e9),
            'as_two_passes_ms': sec_steps * 1e3,
            'note': 'torbi_amd.from_probabilities(probabilities[B,T,S] on the device, batch_frames, gpu=0): log(), the epsilon '
                    'round trip and the uniform-transition decode in ONE pass over the observations '
                    '(torbi_hip_viterbi_decode_uniform_probabilities); beside it the log + clamp pass followed by the decode'}
        del probs
        sec, _ = self.timed_decodes(lambda: self.torbi_amd.decode(obs, frames, trans, init, workspace=ws, path='dense'), 2)
        dense_clock = v.delivered_clock_hz(v.scan_stats(ws, B, T, S))
        record('every_cell', sec, B * T, S, 'headline batch, dense (max,+) GEMM forced: every (prev, next) cell evaluated',
               {'forward_path': 'dense', 'kernel': v.last_forward_kernel(), 'valu_frac_at_1p5_instr_per_cell': 1.5 * B * T * S * S / sec / VALU_LANE_OPS,
                'clock_hz_measured': dense_clock, 'cells_per_s': B * T * S * S / sec,
                'ceiling_cells_per_s_at_measured_clock': valu_ceiling(dense_clock),
                'frac_of_ceiling_at_measured_clock': B * T * S * S / sec / valu_ceiling(dense_clock)})
        # posteriorgram-like input: per-frame log_softmax of peaked logits clamped at log(tiny), the reference's
        # banded pitch transition (torbi/evaluate/core.py:23-34); AUTO's choice after it has settled
        gen = torch.Generator(device=dev).manual_seed(7)
        logits = torch.randn((B, T, S), device=dev, generator=gen) * 2.0
        centre = torch.randint(0, S, (B, T, 1), device=dev, generator=gen)
        logits -= ((torch.arange(S, device=dev)[None, None, :] - centre).abs().float() / 12.0) ** 2
        peaked = torch.log_softmax(logits, dim=-1).clamp_(min=math.log(torch.finfo(torch.float32).tiny))
        del logits
        band = torch.from_numpy(synth.banded_transition(S, args.half_width)).to(dev)
        finite = int((band != float('-inf')).sum())            # cells per timestep and item the band holds

        def executed(seconds, timesteps, forward_ms=None, space=None):
            """The finite cells of the band over the decode (and over its forward pass), priced against the vector ALU's
            ceiling (add, add, 1/2 max3 = 4 issue cycles per cell and wave) at the clock the kernel was delivered."""
            cells = float(finite) * timesteps
            clock = v.delivered_clock_hz(v.scan_stats(space if space is not None else ws, B, T, S))
            found = {'finite_cells_per_timestep': finite, 'fraction_of_all_cells': finite / float(S * S),
                     'cells_per_s': cells / seconds, 'clock_hz_measured': clock,
                     'ceiling_cells_per_s_at_measured_clock': valu_ceiling(clock),
                     'frac_of_ceiling_at_measured_clock': cells / seconds / valu_ceiling(clock)}
            if forward_ms:
                found['forward_cells_per_s'] = cells / (forward_ms * 1e-3)
                found['forward_frac_of_ceiling_at_measured_clock'] = cells / (forward_ms * 1e-3) / valu_ceiling(clock)
            return found

        for _ in range(4):
            self.torbi_amd.decode(peaked, frames, band, init, workspace=ws)
        prof = []
        self.torbi_amd.decode(peaked, frames, band, init, workspace=ws, _profile=prof)
        kernel = v.last_forward_kernel()
        sec, _ = self.timed_decodes(lambda: self.torbi_amd.decode(peaked, frames, band, init, workspace=ws), 3, warmup=0)
        record('peaked_banded', sec, B * T, S,
               'posteriorgram-like rows (log_softmax of peaked logits, clamped at log tiny) with the reference\'s banded '
               'pitch transition (torbi/evaluate/core.py:24-33); path = what AUTO settled on',
               {'forward_path': ROUTES[int(prof[3])], 'kernel': kernel, 'forward_ms': prof[0], 'backtrace_ms': prof[1],
                'us_per_timestep_forward': (prof[0] - prof[4]) * 1e3 / max(T - 1, 1),
                'executed': executed(sec, B * T, prof[0]),
                'statistics_gave_up': int(v.scan_stats(ws, B, T, S).cpu()[127])})
        sec, _ = self.timed_decodes(lambda: self.torbi_amd.decode(peaked, frames, band, init, workspace=ws, path='dense'), 2)
        record('peaked_banded_dense_kernel', sec, B * T, S, 'the same on the dense kernel\'s -inf chunk skipping (AUTO\'s choice '
               'until round 4)')
        sec, _ = self.timed_decodes(lambda: self.torbi_amd.decode(peaked, frames, trans, init, workspace=ws), 3, warmup=4)
        self.torbi_amd.decode(peaked, frames, trans, init, workspace=ws, _profile=prof)
        record('peaked_dense_transition', sec, B * T, S, 'the same peaked rows with the dense random transition (AUTO)',
               {'forward_path': ROUTES[int(prof[3])], 'kernel': v.last_forward_kernel()})
        # the same workload as a launch group of 8 batches (what from_files_to_files sees): AUTO's choice for the group
        spaces = [torch.empty(v.workspace_bytes(B, T, S), dtype=torch.uint8, device=dev) for _ in range(8)]
        prof = []
        v.decode_batches([peaked] * 8, [frames] * 8, band, init, workspaces=spaces, _profile=prof)
        sec, _ = self.timed_decodes(lambda: v.decode_batches([peaked] * 8, [frames] * 8, band, init, workspaces=spaces), 2)
        record('peaked_banded_launch_group', sec, 8 * B * T, S,
               'eight batches of the peaked rows + banded pitch transition in one call',
               {'forward_path': ROUTES[int(prof[3])], 'kernel': v.last_forward_kernel(), 'forward_ms': prof[0],
                'backtrace_ms': prof[1], 'us_per_timestep_forward': (prof[0] - prof[4]) * 1e3 / max(T - 1, 1),
                'members_per_tile': int(self.torbi_amd._lib.load().torbi_hip_band_members(8 * B, S, *v.band_reach(band, band, S),
                                                                                         dev.index or 0)),
                'executed': executed(sec, 8 * B * T, prof[0], spaces[0]),
                'statistics_gave_up': int(v.scan_stats(spaces[0], B, T, S).cpu()[127])})
        sec, _ = self.timed_decodes(lambda: v.decode_batches([peaked] * 8, [frames] * 8, band, init, workspaces=spaces,
                                                             path='resident'), 2)
        record('peaked_banded_launch_group_resident', sec, 8 * B * T, S, 'the same, time-resident kernel forced')
        # ... and with the matrix the reference's EVALUATION really decodes with: from_files_to_files(transition_file=...,
        # log_probs=True) takes log(p + tiny) (torbi/core.py:341-347): log(tiny) = -87.34 outside the band, not -inf
        evaluated = torch.from_numpy(synth.banded_transition(S, args.half_width, tiny=True)).to(dev)
        prof = []
        v.decode_batches([peaked] * 8, [frames] * 8, evaluated, init, workspaces=spaces, _profile=prof)
        sec, _ = self.timed_decodes(lambda: v.decode_batches([peaked] * 8, [frames] * 8, evaluated, init, workspaces=spaces), 2)
        record('evaluation_matrix_launch_group', sec, 8 * B * T, S,
               'eight batches of the peaked rows with log(p + tiny) of the pitch transition, as torbi.evaluate calls '
               'from_files_to_files (torbi/evaluate/core.py:97-103): ONE constant outside the band -- the whole-tile band kernel '
               'decides every output from the band and the row maximum (csrc/band_tile_forward.hpp); AUTO',
               {'forward_path': ROUTES[int(prof[3])], 'kernel': v.last_forward_kernel(), 'forward_ms': prof[0],
                'backtrace_ms': prof[1], 'background': float(evaluated[0, S - 1]),
                'executed': executed(sec, 8 * B * T, prof[0], spaces[0]),
                'statistics_gave_up': int(v.scan_stats(spaces[0], B, T, S).cpu()[127])})
        sec, _ = self.timed_decodes(lambda: v.decode_batches([peaked] * 8, [frames] * 8, evaluated, init, workspaces=spaces,
                                                             path='resident'), 2)
        record('evaluation_matrix_launch_group_resident', sec, 8 * B * T, S, 'the same, time-resident kernel forced (AUTO until round 6)')
        for _ in range(3):
            self.torbi_amd.decode(peaked, frames, evaluated, init, workspace=spaces[0])
        prof = []
        self.torbi_amd.decode(peaked, frames, evaluated, init, workspace=spaces[0], _profile=prof)
        kernel = v.last_forward_kernel()
        sec, _ = self.timed_decodes(lambda: self.torbi_amd.decode(peaked, frames, evaluated, init, workspace=spaces[0]), 3, warmup=0)
        record('evaluation_matrix', sec, B * T, S,
               'ONE batch of the peaked rows with log(p + tiny) of the pitch transition: tiles split over members that exchange '
               'their rows\' maxima (csrc/band_forward.hpp, <true>); AUTO',
               {'forward_path': ROUTES[int(prof[3])], 'kernel': kernel, 'forward_ms': prof[0], 'backtrace_ms': prof[1],
                'executed': executed(sec, B * T, prof[0], spaces[0])})
        del evaluated
        del peaked, band, spaces
        c = float(torch.tensor(math.log(1.0 / S), dtype=torch.float32))
        sec, _ = self.timed_decodes(lambda: self.torbi_amd.decode_uniform(obs, frames, c, init), 3)
        out['uniform'] = {'value': B * T / sec, 'unit': 'timesteps/s', 'ms_per_decode': sec * 1e3,
                          'roofline_frac': B * T * (4 * S + 4) / sec / (HBM_PEAK_GBS * 1e9),
                          'forward_path': 'uniform', 'kernel': 'uniform::uniform_rows_kernel',
                          'note': 'uniform transition (reference default, transition=None): O(S) per timestep, '
                                  'HBM-bound on the 4S observation bytes'}
        del ws
        # BASELINE configs[1]: B = 1
        o1 = obs[:1].contiguous()
        f1 = frames[:1].contiguous()
        sec, _ = self.timed_decodes(lambda: self.torbi_amd.decode(o1, f1, trans, init), 3)
        self.torbi_amd.decode(o1, f1, trans, init, _profile=prof)
        record('c2', sec, T, S, f'BASELINE configs[1]: {S} states, {T} frames, batch=1 (latency bound): ONE launch, the '
                                'matrix held in registers across the chip, posterior rows handed from workgroup to '
                                'workgroup as {value, timestep} words',
               {'us_per_timestep': sec / max(T - 1, 1) * 1e6, 'forward_path': ROUTES[int(prof[3])], 'kernel': v.last_forward_kernel()})
        sec, _ = self.timed_decodes(lambda: self.torbi_amd.decode(o1, f1, trans, init, path='dense'), 3)
        record('c2_per_timestep_kernels', sec, T, S, 'the same with one launch per timestep (what AUTO took before round 3)',
               {'us_per_timestep': sec / max(T - 1, 1) * 1e6})
        # state counts below the reference's pitch bins (a 3-state toy like BASELINE configs[0], 40 classes, 256 bins):
        # one wavefront / workgroup per sequence, the matrix in registers (csrc/small_states.hpp): up to 64 states recurrence
        # and walk back in ONE launch; 65 .. 256 states a value-only forward launch + the backtrace in speculative segments
        for Bs, Ss in ((1, 3), (512, 40), (512, 256), (4096, 64), (512, 128)):
            os_ = v.fill_synthetic((Bs, T, Ss), synth.STREAM_OBSERVATION, seed=7, device=dev)
            ts_ = v.fill_synthetic((Ss, Ss), synth.STREAM_TRANSITION, seed=7, device=dev)
            is_ = v.fill_synthetic((Ss,), synth.STREAM_INITIAL, seed=7, device=dev)
            fs_ = torch.full((Bs,), T, dtype=torch.int32, device=dev)
            self.torbi_amd.decode(os_, fs_, ts_, is_, _profile=prof)
            sec, _ = self.timed_decodes(lambda: self.torbi_amd.decode(os_, fs_, ts_, is_), 3)
            extra = {'us_per_timestep': sec / max(T - 1, 1) * 1e6, 'forward_path': ROUTES[int(prof[3])],
                     'launches': int(prof[2]), 'kernel': v.last_forward_kernel()}
            if Bs * T <= 512 * 500:       # (the per-timestep kernels beside it; not for the large batch: seconds)
                sec_steps, _ = self.timed_decodes(lambda: self.torbi_amd.decode(os_, fs_, ts_, is_, path='dense'), 2)
                extra['one_launch_per_timestep_ms'] = sec_steps * 1e3
            record(f'small_states_{Bs}x{T}x{Ss}', sec, Bs * T, Ss, 'up to 256 states: a wavefront / workgroup per sequence, the '
                                                                   'matrix in registers (`launches` says how many kernels the decode '
                                                                   'took); beside it the per-timestep kernels (DENSE named)',
                   extra)
            del os_, ts_, is_, fs_
        # BASELINE configs[4]: 4096 states, 2000 frames, batch 128
        try:
            B5, T5, S5 = 128, 2000, 4096
            o5 = v.fill_synthetic((B5, T5, S5), synth.STREAM_OBSERVATION, seed=5, device=dev)
            t5 = v.fill_synthetic((S5, S5), synth.STREAM_TRANSITION, seed=0, device=dev)
            i5 = v.fill_synthetic((S5,), synth.STREAM_INITIAL, seed=0, device=dev)
            f5 = torch.full((B5,), T5, dtype=torch.int32, device=dev)
            w5 = torch.empty(v.workspace_bytes(B5, T5, S5), dtype=torch.uint8, device=dev)
            prof = []
            self.torbi_amd.decode(o5, f5, t5, i5, workspace=w5, _profile=prof)
            sec, _ = self.timed_decodes(lambda: self.torbi_amd.decode(o5, f5, t5, i5, workspace=w5), 2)
            record('c5', sec, B5 * T5, S5, 'BASELINE configs[4]: 4096 states, 2000 frames, batch=128 (AUTO: ONE time-resident '
                                           'launch, 16 tiles of 8 items x 16 workgroups each)',
                   {'forward_path': ROUTES[int(prof[3])], 'kernel': v.last_forward_kernel()})
            # the same shape as a launch group (a many-file job at 4096 states): four batches in one time-resident launch
            T5g, n5 = 500, 4
            spaces5 = [torch.empty(v.workspace_bytes(B5, T5g, S5), dtype=torch.uint8, device=dev) for _ in range(n5)]
            obs5 = [o5[:, k * T5g:(k + 1) * T5g].contiguous() for k in range(n5)]
            fr5 = [torch.full((B5,), T5g, dtype=torch.int32, device=dev)] * n5
            prof = []
            v.decode_batches(obs5, fr5, t5, i5, workspaces=spaces5, _profile=prof)
            sec, _ = self.timed_decodes(lambda: v.decode_batches(obs5, fr5, t5, i5, workspaces=spaces5), 2)
            record('c5_shape_launch_group', sec, n5 * B5 * T5g, S5,
                   f'{n5} batches of 128 x {T5g} x 4096 in one call (launch group)', {'forward_path': ROUTES[int(prof[3])]})
            del o5, t5, i5, f5, w5, spaces5, obs5
        except RuntimeError as exc:     # out of memory next to the headline buffers: say so instead of dying
            out['c5'] = {'value': None, 'note': f'not measured: {exc}'}
        out['chunked_long_sequence'] = self.chunked_long_sequence(trans, init)
        try:
            # the CPU operator gpu=None callers get (include/torbi_cpu.h): 64 items of the headline batch on the host's cores
            n = min(64, B)
            host = [x.cpu() for x in (obs[:n], frames[:n], trans, init)]
            t0 = time.perf_counter()
            cpu_indices = self.torbi_amd.decode_cpu(*host)
            sec = time.perf_counter() - t0
            gpu_indices = self.torbi_amd.decode(obs[:n].contiguous(), frames[:n].contiguous(), trans, init).cpu()
            out['cpu_twin'] = {'value': n * T / sec, 'unit': 'timesteps/s', 'seconds': sec, 'items': n,
                               'threads': 'OpenMP default', 'equals_gpu_indices': bool(torch.equal(cpu_indices, gpu_indices)),
                               'note': 'torbi_amd.decode_cpu, the host twin of the operator that gpu=None selects (not '
                                       'the cpu_baseline, which times the reference operator)'}
        except (OSError, RuntimeError) as exc:
            out['cpu_twin'] = {'value': None, 'note': f'not measured: {exc}'}
        try:
            # configs[3] from files to files (4 096 sequences: torch.save()d inputs in /dev/shm -> one output file each)
            out['c4_files_end_to_end'] = self.c4_end_to_end(4096)
            # the reference's DEFAULT call: files of probabilities, log_probs=False -- staged like the others, log() and the
            # epsilon round trip as one pass in place in the device slab (a smaller job: the files are written once more)
            record = self.c4_end_to_end(1024, probabilities=True)
            out['c4_files_of_probabilities'] = {k: record[k] for k in ('value', 'unit', 'seconds', 'first_call_seconds', 'sequences',
                                                                        'files_hold', 'gb_per_s_from_files')}
        except (OSError, RuntimeError) as exc:
            out['c4_files_end_to_end'] = {'value': None, 'note': f'not measured: {exc}'}
        return out

    def chunked_long_sequence(self, trans, init):
        """one long sequence (B = 1 is latency bound: ~4.7 us per frame whatever the GPU) cut at
        low-entropy frames (torbi_amd/chunk.py = reference torbi/chunk.py) into pieces that are decoded as batch rows
        and joined.  An approximation (results may differ from the unchunked decode at the cuts); reported: frames/s
        both ways and the fraction of frames on which the two agree."""
        torch = self.torch
        import torbi_amd
        from torbi_amd import data as tdata
        dev, S, T = self.dev, trans.shape[0], 16000
        gen = torch.Generator().manual_seed(11)
        logits = torch.randn(T, S, generator=gen) * 2.0
        centre = (S / 2 + S / 3 * torch.sin(torch.arange(T) / 150.0)).long().clamp(0, S - 1)
        logits -= ((torch.arange(S)[None, :] - centre[:, None]).abs().float() / 10.0) ** 2
        certain = (torch.arange(T) % 97) < 2                      # two adjacent near-certain frames every 97 frames
        logits[certain, centre[certain]] += 60.0
        sequence = torch.log_softmax(logits, dim=-1).clamp_(min=math.log(torch.finfo(torch.float32).tiny))
        band = torch.from_numpy(self.synth.banded_transition(S, self.args.half_width)).to(dev)
        whole = sequence[None].to(dev)
        frames = torch.tensor([T], dtype=torch.int32, device=dev)
        sec_whole, idx_whole = self.timed_decodes(lambda: torbi_amd.decode(whole, frames, band, init), 2)
        t0 = time.perf_counter()
        pieces = torbi_amd.chunk(sequence, min_chunk_size=200, entropy_threshold=0.5)
        observation, batch_frames, batch_chunks, _ = tdata.collate([(pieces, 'sequence')])
        cut_s = time.perf_counter() - t0
        observation = observation.to(dev)
        batch_frames = batch_frames.to(device=dev, dtype=torch.int32)
        sec_chunks, idx_rows = self.timed_decodes(lambda: torbi_amd.decode(observation, batch_frames, band, init), 3)
        joined = tdata.separate(idx_rows.cpu(), batch_chunks, batch_frames.cpu())[0]
        agree = float((joined == idx_whole[0].cpu()).float().mean())
        return {'value': T / sec_chunks, 'unit': 'timesteps/s', 'unchunked_value': T / sec_whole,
                'ms_chunked': sec_chunks * 1e3, 'ms_unchunked': sec_whole * 1e3, 'pieces': len(pieces),
                'longest_piece': int(batch_frames.max()), 'host_entropy_and_cut_ms': cut_s * 1e3,
                'frames_agreeing_with_unchunked': agree,
                'note': f'one sequence of {T} frames x {S} states, banded pitch transition, MIN_CHUNK_SIZE = 200: decode '
                        f'time only (the cut points are found on the host like upstream, dataset.py:22-23)'}

    # ---- BASELINE configs[3]: ragged many-file job ----------------------------------------------------
    def c4_plan(self, files):
        """Batches of the job in file order (reference loader: BATCH_SIZE files per batch, collate pads to the batch
        maximum) and their assignment to ranks by padded cost."""
        lengths = self.synth.lengths(files, 100, 900).tolist()
        plan = self.distributed.assign_batches(lengths, self.args.batch, self.size)
        return lengths, plan

    def c4_decode_only(self, files, steps, quiet=False):
        args, torch, v, synth = self.args, self.torch, self.viterbi, self.synth
        S = args.states
        lengths, plan = self.c4_plan(files)
        mine = plan[self.rank] if self.size > 1 else [b for r in plan for b in r]
        mine = sorted(mine, key=lambda b: b[0])
        if steps:
            mine = mine[:steps]
        valid = sum(lengths[i] for b in mine for i in b)
        padded = sum(max(lengths[i] for i in b) * len(b) for b in mine)
        if self.dry:
            self.fence()
            total = self.sum_over_ranks(valid)
            return {'value': None, 'dry_run': True, 'batches_per_rank': [len(r) for r in plan],
                    'valid_timesteps': total}
        dev = self.dev
        trans, init = self.model(S)
        group = max(1, args.group)
        slots = group * max(1, args.pipeline)
        tmax = max((max(lengths[i] for i in b) for b in mine), default=1)
        # a pool of resident observation buffers: batch k decodes buffer k % slots with ITS lengths
        pool = [v.fill_synthetic((args.batch, tmax, S), synth.STREAM_OBSERVATION, seed=self.rank * 64 + k, device=dev)
                for k in range(min(slots, max(len(mine), 1)))]
        batches = []
        for k, b in enumerate(mine):
            t = max(lengths[i] for i in b)
            frames = torch.tensor([lengths[i] for i in b], dtype=torch.int32, device=dev)
            # a contiguous (items, longest, S) batch laid over the head of the pool buffer (synthetic values: any
            # reinterpretation of the buffer is as good a batch as any other)
            flat = pool[k % len(pool)].view(-1)
            batches.append((flat[:len(b) * t * S].view(len(b), t, S), frames))
        pipe = self.torbi_amd.DecodePipeline(dev, depth=max(1, args.pipeline), group=group,
                                             path=None if args.forward == 'auto' else args.forward)
        pipe.reserve(args.batch, tmax, S)

        def run():
            for observation, frames in batches:
                pipe.decode(observation, frames, trans, init)
            pipe.synchronize()

        run()                                   # warm-up pass over the same batches
        self.fence(pipe)
        t0 = time.perf_counter()
        run()
        self.fence(pipe)
        elapsed = self.max_over_ranks(time.perf_counter() - t0)
        total_valid = self.sum_over_ranks(valid)
        total_padded = self.sum_over_ranks(padded)
        record = {'value': total_valid / elapsed, 'unit': 'timesteps/s', 'seconds': elapsed,
                  'sequences': files if not steps else sum(len(b) for b in mine), 'batches': len(mine),
                  'valid_timesteps': total_valid, 'padded_timesteps': total_padded,
                  'padding_overhead': total_padded / max(total_valid, 1.0) - 1.0,
                  'roofline_frac': total_valid / elapsed * algorithmic_bytes_per_timestep(S) / (HBM_PEAK_GBS * 1e9),
                  'note': 'lengths 100..900 in file order, batches of 512 padded to the batch maximum like '
                          'collate.py:24-31; value counts VALID timesteps only; the time-resident kernel runs every '
                          '16-item tile to the longest of ITS items, so padding costs memory, not recurrence steps'}
        del pool, batches
        return record

    def c4_end_to_end(self, files, probabilities=False):
        """torch.save()d inputs -> from_files_to_files -> torch.save()d outputs on this rank's share.  `probabilities`: the
        files hold probabilities and the call is the reference's default (log_probs=False, torbi/core.py:310-318)."""
        import shutil
        import tempfile
        torch, synth = self.torch, self.synth
        S = self.args.states
        lengths = synth.lengths(files, 100, 900).tolist()
        base = '/dev/shm' if os.path.isdir('/dev/shm') else None
        folder = tempfile.mkdtemp(prefix='torbi_c4_', dir=base)
        try:
            ins, outs = [], []
            gen = torch.Generator().manual_seed(1)
            block = torch.rand(900, S, generator=gen)
            block = block.softmax(-1) if probabilities else block.log_softmax(-1)
            for k, n in enumerate(lengths):
                f = os.path.join(folder, f'in{k}.pt')
                if self.rank == 0:
                    torch.save(torch.roll(block, k, dims=0)[:n].clone(), f)
                ins.append(f)
                outs.append(os.path.join(folder, f'out{k}.pt'))
            tf = os.path.join(folder, 'transition.pt')
            if self.rank == 0:
                torch.save(torch.rand(S, S, generator=gen).softmax(-1), tf)
            workers = min(32, max(1, (os.cpu_count() or 2) // (2 * self.size)))
            seconds = []
            # the inputs were written a moment ago: their tmpfs pages are touched for the first time by whoever reads them
            # first.  A separate process reads every file once BEFORE the timed calls, so that first_call_seconds is the
            # library cold (pinned blocks, device scratch, code objects) and not the page cache's first touch.
            preread = False
            if self.rank == 0:
                listing = os.path.join(folder, 'inputs.txt')
                with open(listing, 'w') as fh:
                    fh.write('\n'.join(ins + [tf]))
                os.sync()
                preread = subprocess.run([sys.executable, '-c',
                                          'import sys\nfor f in open(sys.argv[1]).read().split("\\n"):\n    open(f, "rb").read()',
                                          listing], timeout=600).returncode == 0
            core = self.torbi_amd.core
            kept_before, core.KEEP_JOB_MEMORY = core.KEEP_JOB_MEMORY, True     # a long-running job server: see the note
            for _ in range(2):      # first call: pinned host blocks, device scratch and code objects are new
                self.fence()
                t0 = time.perf_counter()
                self.distributed.from_files_to_files(ins, outs, transition_file=tf, log_probs=not probabilities,
                                                     lengths=lengths, num_workers=workers)
                self.fence()
                seconds.append(self.max_over_ranks(time.perf_counter() - t0))
            elapsed = seconds[-1]
            core.KEEP_JOB_MEMORY = kept_before
            core.release_job_memory()
            ok = all(os.path.exists(f) for f in outs)
            direct = bool(getattr(self.torbi_amd.core, 'DIRECT_FILE_IO', False))
            return {'value': sum(lengths) / elapsed, 'unit': 'timesteps/s', 'seconds': elapsed,
                    'first_call_seconds': seconds[0], 'first_call_over_steady': seconds[0] / elapsed,
                    'inputs_read_once_by_another_process_before_timing': preread, 'sequences': files,
                    'outputs_written': ok, 'files_hold': 'probabilities (log_probs=False)' if probabilities else 'log-probabilities',
                    'reader_threads': workers,
                    'host_path': 'direct reader (payloads pread into pinned batch rows by native threads, outputs from '
                                 'a prebuilt container image)' if direct else 'torch.load + collate in DataLoader workers, torch.save',
                    'gb_per_s_from_files': sum(lengths) * S * 4 / elapsed / 1e9,
                    'roofline': {'bound': 'pcie', 'achieved': sum(lengths) * S * 4 / elapsed / 1e9, 'peak': PCIE_PEAK_GBS,
                                 'unit': 'GB/s', 'frac': sum(lengths) * S * 4 / elapsed / 1e9 / PCIE_PEAK_GBS,
                                 'note': 'every frame crosses the host link once as 4 * S bytes of fp32 log-probabilities '
                                         '(MI355X_MICROARCH.md: PCIe Gen5 x16, 63 GB/s spec; 57 GB/s measured for pinned H2D)'},
                    'note': 'files -> pinned batches -> H2D -> epsilon clamp -> decode -> D2H -> one output file per input, '
                            f'length-bucketed batches, files in {folder.rsplit("/", 1)[0]}; value = the second call '
                            '(steady state of a process that runs job after job with torbi_amd.core.KEEP_JOB_MEMORY = True: '
                            'pipeline scratch and staging slabs kept between jobs), first_call_seconds = the same job cold '
                            '(what a single job sees; by default the scratch is freed with the job)'}
        finally:
            if self.rank == 0:
                shutil.rmtree(folder, ignore_errors=True)

    def run_c4(self):
        args = self.args
        record = self.c4_decode_only(args.files, args.steps if args.steps != 16 else 0)
        result = {'metric': 'timesteps decoded/sec, 1440 states, ragged many-file job (BASELINE configs[3])',
                  'value': record.get('value'), 'unit': 'timesteps/s', 'n_gpus': self.size,
                  'steps': record.get('batches'), 'warmup': record.get('batches'),
                  'ms_per_step': (record['seconds'] / max(record['batches'], 1) * 1e3) if record.get('seconds') else None,
                  'higher_is_better': True, 'scaling': 'strong', 'vs_baseline': None, 'dtype': 'f32',
                  'data': 'synthetic',
                  'config': {'workload': f'{args.states} states, {args.files} sequences of 100..900 frames, batches of '
                                         f'{args.batch} padded to the batch maximum, decode only, inputs resident in HBM; '
                                         f'batches assigned to ranks by padded cost',
                             'parallelism': f'batches sharded x{self.size}' if self.size > 1 else 'single GPU',
                             'launch_group': args.group, 'streams': args.pipeline},
                  'decode_only': record}
        if record.get('dry_run'):
            result['dry_run'] = True
        if args.end_to_end and not self.dry:
            result['end_to_end'] = self.c4_end_to_end(args.end_to_end)
        return result


def main():
    args = parse_args()
    if args.gpus > 1 and 'WORLD_SIZE' not in os.environ:
        sys.exit(self_launch(args))
    bench = Bench(args)
    result = bench.run_c4() if args.workload == 'c4' else bench.run_c3()
    if bench.collective:
        bench.dist.destroy_process_group()
    # RCCL announces itself through C stdio ("Librccl path : ..."), which would otherwise be flushed at exit, AFTER
    # the result: flush it now so that the JSON line is the last thing on stdout
    try:
        import ctypes
        ctypes.CDLL(None).fflush(None)
    except OSError:
        pass
    sys.stdout.flush()
    if bench.rank == 0:
        print(json.dumps(result), flush=True)


if __name__ == '__main__':
    main()
