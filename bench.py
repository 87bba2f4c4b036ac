"""Headline benchmark: timesteps decoded / second, 1440 states, batch 512 (BASELINE.json).

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

A "step" is ONE decode (forward recurrence + final argmax + backtrace) of a synthetic batch
of 512 sequences x 500 frames x 1440 states (BASELINE configs[2]) that is already resident
in HBM.  With N GPUs every rank decodes its own such batch (weak scaling, batch items are
independent) and the decoded indices are all-gathered over RCCL inside the timed region.
Rank 0 prints one JSON line; see DESIGN.md "Measurement" for every field.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import numpy as np
import torch
import torch.distributed as dist

import torbi_amd
from torbi_amd import distributed, synth, viterbi

HBM_PEAK_GBS = 8000.0                 # MI355X_MICROARCH.md: 8.0 TB/s spec
VALU_LANE_OPS = 256 * 4 * 32 * 2.4e9  # 256 CUs x 4 SIMD-32 x 2.4 GHz lane-instructions/s


def profiled_traffic(kernel_prefix):
    """HBM-side bytes per launch of the dominant kernel from the committed rocprofv3 PMC summary
    (profiles/r01_pmc.json: separate --pmc FETCH_SIZE / WRITE_SIZE passes of this same command).
    FETCH_SIZE is doubled as MI355X_MICROARCH.md prescribes for wide coalesced reads on gfx950;
    both counters are in KiB and include Infinity-Cache hits.  None when no summary matches."""
    try:
        pmc = json.load(open(os.path.join(ROOT, 'profiles', 'r01_pmc.json')))
        for name, counters in pmc.items():
            if kernel_prefix in name and 'FETCH_SIZE' in counters and 'WRITE_SIZE' in counters:
                return (2.0 * counters['FETCH_SIZE']['mean_per_dispatch']
                        + counters['WRITE_SIZE']['mean_per_dispatch']) * 1024.0
    except (OSError, ValueError, KeyError):
        pass
    return None


def algorithmic_bytes_per_timestep(S):
    """4S observation read + 4S int32 backpointer write + 8 (backtrace)."""
    return 8 * S + 8


def cpu_baseline(obs_dev, trans_dev, init_dev, gpu_indices, budget_s=15.0):
    """Time the CPU path on this host's cores on a bounded sample of the SAME workload and
    check the GPU's indices against it.  Uses the reference's own operator (oracle/_ref,
    kind "reference") when that build is present, else the reference-shaped C port (oracle
    mode 0, kind "port").  The thread count is calibrated on a 16-frame prefix (the
    reference's at::parallel_for over states gets SLOWER with hundreds of threads) and the
    sample is sized to about `budget_s` seconds."""
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
    best_threads, best_dt = 1, None
    for threads in (1, 2, 4, 8, 16, 32, 64, 128):
        if threads > host:
            break
        dt, _ = run(cal, np.array([16], np.int32), threads)
        if best_dt is None or dt < best_dt:
            best_threads, best_dt = threads, dt
        elif dt > 2 * best_dt:
            break
    per_item = best_dt / 15.0 * (T - 1)            # 15 recurrence steps in the prefix
    items = int(max(1, min(obs_dev.shape[0], budget_s / max(per_item, 1e-3))))
    obs = obs_dev[:items].cpu().numpy()
    dt, idx = run(obs, np.full((items,), T, np.int32), best_threads)
    match = bool(np.array_equal(idx, gpu_indices[:items].cpu().numpy()))
    return {
        'value': items * T / dt, 'unit': 'timesteps/s', 'cores': best_threads,
        'kind': 'reference' if use_ref else 'port',
        'sample': f'first {items} of {obs_dev.shape[0]} items x {T} frames x {S} states, '
                  f'{best_threads} threads (best of a 16-frame calibration; host has {host} '
                  f'logical CPUs), {dt:.1f} s',
        'gpu_matches_cpu': match,
    }


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=10)
    ap.add_argument('--warmup', type=int, default=2)
    ap.add_argument('--batch', type=int, default=512)
    ap.add_argument('--frames', type=int, default=500)
    ap.add_argument('--states', type=int, default=1440)
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--reuse-preparation', action='store_true',
                    help='let consecutive decodes share the per-transition preparation (sorted rows / packed '
                         'panels) as a serving loop would; off by default: every timed decode does all of its work')
    ap.add_argument('--forward', choices=['auto', 'dense', 'pruned'], default='auto',
                    help='forward-recurrence path (include/torbi_hip.h); every path gives identical indices')
    ap.add_argument('--pipeline', type=int, default=2,
                    help='decodes in flight (torbi_amd.DecodePipeline streams); 1 = strictly serial')
    ap.add_argument('--transition', choices=['dense', 'banded', 'uniform'], default='dense',
                    help="dense = headline workload; banded = the reference's pitch transition "
                         '(torbi/evaluate/core.py:24-33), secondary structured-transition line')
    ap.add_argument('--half-width', type=float, default=87.2,
                    help='band half width in states for --transition banded (penn: 87.2)')
    args = ap.parse_args()

    rank, size, local = distributed.init_from_env()
    assert size == args.gpus, f'--gpus {args.gpus} but WORLD_SIZE={size}'
    dev = torch.device('cuda', torch.cuda.current_device())
    collective = dist.is_available() and dist.is_initialized()
    B, T, S = args.batch, args.frames, args.states
    viterbi.set_forward_path(args.forward)

    # synthetic inputs generated in HBM (rank-specific observation stream; shared transition)
    obs = viterbi.fill_synthetic((B, T, S), synth.STREAM_OBSERVATION, seed=rank, device=dev)
    trans = viterbi.fill_synthetic((S, S), synth.STREAM_TRANSITION, seed=0, device=dev)
    if args.transition == 'banded':
        trans = torch.from_numpy(synth.banded_transition(S, args.half_width)).to(dev)
    init = viterbi.fill_synthetic((S,), synth.STREAM_INITIAL, seed=0, device=dev)
    frames = torch.full((B,), T, dtype=torch.int32, device=dev)
    ws = torch.empty(viterbi.workspace_bytes(B, T, S), dtype=torch.uint8, device=dev)

    import math
    uniform_c = float(torch.tensor(math.log(1.0 / S), dtype=torch.float32))

    pipe = torbi_amd.DecodePipeline(dev, depth=args.pipeline, reuse_preparation=args.reuse_preparation) \
        if args.pipeline > 1 else None

    def gather(idx):
        return distributed.gather_indices(idx, B * size, force=True) if collective else idx

    def step():
        if args.transition == 'uniform':
            return gather(torbi_amd.decode_uniform(obs, frames, uniform_c, init))
        if pipe is not None:     # consecutive batches alternate between HIP streams
            return pipe.decode(obs, frames, trans, init, after=gather)
        return gather(torbi_amd.decode(obs, frames, trans, init, workspace=ws,
                                       reuse_preparation=args.reuse_preparation))

    def fence():
        if pipe is not None:
            pipe.synchronize()
        torch.cuda.synchronize()
        if collective:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    fence()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        indices = step()
    fence()
    elapsed = time.perf_counter() - t0
    if collective:
        tmax = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed = float(tmax.item())

    timesteps = float(B) * T * args.steps * size
    value = timesteps / elapsed

    # dominant kernel: the forward-recurrence step kernel; average launch duration measured with
    # hipEvents on the launch stream around the whole chain of launches (one launch = one
    # timestep of the whole batch), averaged over a few profiled decodes
    prof, fwd_ms, bt_ms, launches = [], 0.0, 0.0, 1
    if args.transition == 'uniform':
        # one kernel per decode, HBM-bound: 4S observation bytes + 4 index bytes per timestep
        per = elapsed / args.steps
        nbytes = float(B) * T * (4 * S + 4)
        result = {
            'metric': 'timesteps decoded/sec, 1440 states batch=512', 'value': value,
            'unit': 'timesteps/s', 'n_gpus': size, 'steps': args.steps, 'warmup': args.warmup,
            'ms_per_step': per * 1e3, 'higher_is_better': True, 'scaling': 'weak',
            'vs_baseline': None, 'dtype': 'f32', 'data': 'synthetic',
            'config': {'workload': f'{S} states, {T} frames, batch={B} per GPU, fp32, UNIFORM '
                                   f'transition (the reference default, transition=None; secondary '
                                   f'workload)'},
            'roofline': {'bound': 'hbm', 'achieved': nbytes / per / 1e9, 'peak': HBM_PEAK_GBS,
                         'unit': 'GB/s', 'frac': nbytes / per / 1e9 / HBM_PEAK_GBS, 'traffic': None,
                         'kernel': 'uniform_decode_kernel (whole decode, host-timed incl. launch)',
                         'algorithmic_bytes_per_launch': nbytes},
        }
        if rank == 0:
            print(json.dumps(result), flush=True)
        if collective:
            dist.destroy_process_group()
        return
    for _ in range(3):
        torbi_amd.decode(obs, frames, trans, init, workspace=ws, _profile=prof)
        fwd_ms += prof[0] / 3
        bt_ms += prof[1] / 3
        launches = max(int(prof[2]), 1)
    path = {2: 'pruned', 1: 'dense', 0: 'generic'}[int(prof[3])]
    step_kernel = {'pruned': 'pruned::step_pruned_kernel<16, false>', 'dense': 'step_dense_kernel<8, 6, 8, 12>',
                   'generic': 'step_rows_kernel'}[path]
    per_launch_s = fwd_ms * 1e-3 / launches
    bytes_per_launch = B * algorithmic_bytes_per_timestep(S)
    achieved = bytes_per_launch / per_launch_s / 1e9
    cells_per_launch = float(B) * S * S
    result = {
        'metric': 'timesteps decoded/sec, 1440 states batch=512',
        'value': value, 'unit': 'timesteps/s', 'n_gpus': size, 'steps': args.steps,
        'warmup': args.warmup, 'ms_per_step': elapsed / args.steps * 1e3,
        'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None, 'dtype': 'f32',
        'data': 'synthetic',
        'config': {'workload': f'{S} states, {T} frames, batch={B} per GPU, fp32, dense '
                               f'transition{" (BASELINE configs[2])" if (B, T, S) == (512, 500, 1440) else ""}; '
                               f'decode = forward + argmax + backtrace, inputs resident in HBM'
                               if args.transition == 'dense' else
                               f'{S} states, {T} frames, batch={B} per GPU, fp32, BANDED transition '
                               f'(half width {args.half_width}, -inf outside; secondary workload)',
                   'parallelism': f'batch-sharded x{size}' if size > 1 else 'single GPU',
                   'decodes_in_flight': args.pipeline, 'forward_path': path,
                   'transition_preparation': 'reused across decodes' if args.reuse_preparation
                   else 'rebuilt by every decode'},
        'roofline': {
            'bound': 'hbm', 'achieved': achieved, 'peak': HBM_PEAK_GBS, 'unit': 'GB/s',
            'frac': achieved / HBM_PEAK_GBS,
            'traffic': profiled_traffic(step_kernel)
            if (B, T, S, args.transition) == (512, 500, 1440, 'dense') else None,
            'traffic_note': 'bytes per launch from profiles/r01_pmc.json (2*FETCH_SIZE + WRITE_SIZE, '
                            'Infinity-Cache hits included); the excess over the algorithmic bytes is '
                            'posterior rows / transition lists re-read by the tiles of one launch',
            'kernel': f'{step_kernel} (forward step: one timestep of the whole batch per launch)',
            'launch_us': per_launch_s * 1e6, 'launches_per_decode': launches,
            'algorithmic_bytes_per_launch': bytes_per_launch,
            'note': 'the (max,+) recurrence holds S/4 = 360 cells per algorithmic byte: the dense kernel is '
                    'VALU-bound, the pruned kernel is bound by the CU load path (list entries through the '
                    'texture path, posterior gathers through the LDS); see valu and DESIGN.md',
        },
        'valu': {
            # cells of the full S x S recurrence per second: for the pruned path most are never evaluated
            'dense_equivalent_cells_per_s': cells_per_launch / per_launch_s,
            'lane_instr_peak_per_s': VALU_LANE_OPS,
            'frac_at_1_instr_per_cell': cells_per_launch / per_launch_s / VALU_LANE_OPS,
        },
        'phases_ms': {'forward': fwd_ms, 'argmax_backtrace': bt_ms},
        'hbm_roofline_frac_whole_decode':
            value / size * algorithmic_bytes_per_timestep(S) / (HBM_PEAK_GBS * 1e9),
    }
    if rank == 0 and size == 1 and not args.no_cpu_baseline:
        result['cpu_baseline'] = cpu_baseline(obs, trans, init, indices)
    if rank == 0:
        print(json.dumps(result), flush=True)
    if collective:
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
