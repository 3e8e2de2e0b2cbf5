#!/usr/bin/env python3
"""bench.py -- LM iterations/s and Jacobian-sweep HBM GB/s of the MI355X hot path.

Workload (BASELINE.json metric / SURVEY.md §8d cfg 5): 8 skewed Gaussians = 32 active
parameters, N = 1e7 synthetic points PER GPU (weak scaling: 1e7 x n_gpus points sharded by
the reference's contiguous partition rule, gadfit.F90:977-983), fp64, sigma given (USER).

One "step" = one LM iteration of gfh_fit, the library's gadf_fit main loop (gadfit.F90:671-917):
  STEP 1+2: residual/Jacobian sweep fused with J^T J / J^T r / sum r^2 on the matrix cores ->
  sum over ranks (RCCL all-reduce) -> damped solve on the host (potr) -> parameter update ->
  chi2 at the trial parameters (+ its all-reduce) -> accept/reject and lambda update.
The timed region runs whole fits of FIT_ITERS iterations each (max_iter exit, no other
convergence test), every fit from the same 5 %-off start values, so the K timed iterations are
iterations of a fit in progress (accepted steps) and not of a converged one.  Two schedules
are timed back to back, K iterations each:
  * look-ahead (library default, `value`): the trial chi2 is the sum r^2 the fused sweep returns
    at the trial point, and an accepted step hands that sweep to the next iteration -- one
    N-sized pass per accepted iteration (gadfit_hip.h, gfh_set_lookahead);
  * reference schedule (`reference_schedule`): chi2() kernel at the trial point, then the sweep
    of the same point in the next iteration, exactly the reference's sequence of passes.
Further legs, reported beside `value` and never as `value`: the look-ahead schedule without the
Jacobian store (`jacobian_not_kept`), with geodesic acceleration (`accelerated_fit`), the first iterations after
an idle gap (`cold_start`), and `setup`: host-clock milliseconds of every step from a fresh context to the end of a
first fit, and of the first `gadf_fit` of the same workload through the Fortran API with its phases.
Inputs are resident in HBM before the timed region.  `value` = data points x LM iterations
per second over the whole job; `lm_iters_per_s` is the same thing per iteration.
Steady state: after an idle gap the part's power management slows launches ~3-40 of a back-to-back
series by up to 35 % (tools/probes/transient.py), so PRE_ROLL untimed iterations run before the first
timed leg, on top of --warmup; the timed region is exactly K iterations (`config.pre_roll_steps`).

  python bench.py [--gpus N] [--steps K] [--warmup W] [--points P]
  python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...     (one process per GPU, RCCL)
  python bench.py --gpus N          (no launcher: one process, N member threads of the library's device group, sums by RCCL)
The main leg is repeated (K timed iterations each time, every repeat bracketed by the barrier + synchronize pair) until the
timed regions add up to MIN_TIMED_S; `value` / `ms_per_step` are the MEDIAN repeat, min and max beside them (`repeats`).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # MI355X HBM3E peak (MI355X_MICROARCH.md: 8.0 TB/s spec; ~6.3 achievable)
P_ACTIVE = 32
FIT_ITERS = 10                 # LM iterations per gfh_fit call in the timed region
# After an idle gap the part's power management slows launches ~3-40 of a back-to-back series by up to 35 %
# (tools/probes/transient.py: 0.49 ms -> 0.68 ms -> 0.49 ms from launch ~40 on).  The timed legs measure the steady
# state: this many untimed iterations run first, on top of --warmup.
PRE_ROLL = 64
MIN_TIMED_S = 2.0              # the main leg repeats its K timed iterations until the timed regions add up to this
SWEEP_BYTES_PER_POINT = 24 + 8 + 8 * P_ACTIVE     # read x,y,w; write res and 32 Jacobian entries
GRAM_BYTES_PER_POINT = 8 * P_ACTIVE + 8
CHI2_BYTES_PER_POINT = 24 + 8


PARITY_SUMS_TOL = 1e-13        # all-reduced [JTJ | JTres | chi2] against the rank-ordered host sum of the ranks' partials (scaled, see _sum_deviation)
PARITY_FIT_TOL = 1e-10         # fitted parameters of the N-rank fit against the one-rank fit of the same points (north_star's bound)
PARITY_FIT_POINTS = 200_000
PARITY_FIT_ITERS = 6           # (LM iterations of those fits: further on they have converged to where accept / reject of a trial step is decided by rounding,
                               #  and an N-rank and a one-rank fit may then stop an iteration apart -- SURVEY section 4)
ALLREDUCE_ROUNDS = 300


def _sum_deviation(got, ref):
    """largest deviation of (JTJ, JTres, chi2) `got` from `ref`: JTJ[i,j] in units of sqrt(JTJ_ii JTJ_jj), JTres[i] in units of
    sqrt(JTJ_ii chi2) (the Cauchy-Schwarz sizes of the entries), chi2 relative"""
    import numpy as np
    d = np.sqrt(np.abs(np.diag(ref[0])))
    sc = np.outer(d, d); sc[sc == 0] = 1.0
    sr = d * np.sqrt(abs(ref[2])); sr[sr == 0] = 1.0
    return {'JTJ': float(np.max(np.abs(got[0] - ref[0]) / sc)), 'JTres': float(np.max(np.abs(got[1] - ref[1]) / sr)),
            'chi2': float(abs(got[2] - ref[2]) / abs(ref[2]))}


def _ordered_sum(parts):
    """rows added in rank order, as co_sum's result is defined here (misc.F90:133-170 adds the images' copies one by one)"""
    acc = [parts[0][0].copy(), parts[0][1].copy(), float(parts[0][2])]
    for q in parts[1:]:
        acc[0] += q[0]; acc[1] += q[1]; acc[2] += float(q[2])
    return acc


def _rank_partial(_lib, tape, device, world, r, n_total, begin, x, y, sigma, active, is_global, start, local):
    """this rank's own [JTJ | JTres | chi2] over its share of the points: a second, communicator-less context with the geometry of
    rank r of `world` (gfh_debug_set_rank) on the same card -- nothing is summed across ranks inside it"""
    c = _lib.Context(device)
    try:
        c.debug_set_rank(world, r)
        c.set_model(tape)
        c.set_keep_jacobian(0)                 # (the sums are bitwise the same with and without the Jacobian store)
        if local:
            c.set_data_local(n_total, [0, n_total], begin, x, y, sigma)
        else:
            c.set_data(x, y, sigma, [0, n_total])
        c.init_weights(4)
        jac, dim = c.jacobian_indices(active, is_global)
        return c.sweep(start, active, jac, dim)
    finally:
        c.close()


def dry_line(args):
    """`--dry`: the multi-GPU line's SCHEMA on a device group of compile-only members (devices = -1; runs without a GPU): the
    parity block compares the group's own host sum (gfh_debug_group_allreduce) of synthetic per-rank partials with the rank-ordered
    sum, allreduce_us times that host sum; everything that needs a kernel is null."""
    import numpy as np
    from gadfit_amd import _lib
    n = max(2, args.gpus)
    dim = P_ACTIVE
    g = _lib.Context(devices=[-1] * n)
    rng = np.random.default_rng(5)
    parts = rng.standard_normal((n, dim * dim + dim + 1))
    bufs = parts.copy(); status = np.zeros(n, dtype=np.int32)
    g.debug_group_allreduce(bufs, status)
    ref = parts[0].copy()
    for r in range(1, n):
        ref += parts[r]
    dev = float(np.max(np.abs(bufs - ref[None, :])))
    lat = g.debug_allreduce_latency(dim * dim + dim + 1, 200)
    g.close()
    out = _multi_gpu_block(n, 'host', {'JTJ': dev, 'JTres': dev, 'chi2': dev}, True, None, lat, dim * dim + dim + 1, None, None, None)
    out.update({'dry': True, 'metric': 'LM iterations/s x data points, N=1e7 pts/GPU x 32 active params (whole job)', 'value': None,
                'unit': 'point-iterations/s', 'n_gpus': n, 'steps': args.steps, 'warmup': args.warmup, 'ms_per_step': None,
                'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None, 'dtype': 'f64', 'data': 'synthetic',
                'rccl_nranks': 0, 'config': {'workload': 'dry members (no GPU): schema of the multi-GPU line only'}})
    print(json.dumps(out))


def _multi_gpu_block(world, path, sums_dev, bitwise_ranks, fit, lat, doubles, strong, host_leg, rccl_ms):
    """the keys every N > 1 line carries (tests/test_cpu_bench_schema.py pins them)"""
    sums_ok = sums_dev is not None and max(sums_dev.values()) <= PARITY_SUMS_TOL
    fit_ok = fit is None or bool(fit.get('ok'))
    return {
        'multi_gpu_parity': {
            'ranks': world, 'cross_rank_sum_path': path,
            'sums_vs_ordered_host_sum': {'max_dev': sums_dev, 'tol': PARITY_SUMS_TOL, 'ok': bool(sums_ok),
                                         'what': 'the all-reduced [JTJ | JTres | chi2] of the first sweep at the start parameters against the ranks\' own '
                                                 'partials (communicator-less contexts with the geometry of each rank) added on the host in rank order; '
                                                 'JTJ in units of sqrt(JTJ_ii JTJ_jj), JTres of sqrt(JTJ_ii chi2), chi2 relative',
                                         'all_ranks_hold_the_same_bits': bitwise_ranks},
            'fit_vs_one_rank': fit,
            'ok': bool(sums_ok and fit_ok and bitwise_ranks is not False)},
        'allreduce_us': None if lat is None else {
            'median': lat['median_us'], 'p95': lat['p95_us'], 'min': lat['min_us'], 'max': lat['max_us'],
            'host_round_trip_median': lat['host_round_trip_median_us'], 'doubles': doubles, 'rounds': ALLREDUCE_ROUNDS, 'ranks_counted': lat['nranks'],
            'path': 'ncclAllReduce between two HIP events on the idle stream of rank 0\'s context' if path == 'rccl' else 'ordered host sum of the device group, host clock',
            'note': 'gfh_debug_allreduce_latency: the packed [JTJ | JTres | chi2 | status] image of this workload; host_round_trip = enqueue -> numbers in the host mailbox'},
        'strong_leg': strong,
        'rccl_ms_per_step': rccl_ms,
        'host_sum_ms_per_step': None if host_leg is None else host_leg.get('ms_per_step'),
        'host_sum_leg': host_leg,
    }


from tools.bench_legs import configs_leg, setup_leg, nostore_roofline      # noqa: E402  (the auxiliary legs: tools/bench_legs.py)


def _cpu_share():
    """host cores this process may really use: the affinity mask, cut down to the cgroup's CPU quota where there is one (a GPU
    box hands a job the share of its card, not the machine: 128 workers on a 16-core quota only queue)"""
    try:
        cores = len(os.sched_getaffinity(0))
    except AttributeError:
        cores = os.cpu_count() or 1
    quota = None
    try:
        q, per = open('/sys/fs/cgroup/cpu.max').read().split()[:2]
        if q != 'max':
            quota = float(q) / float(per)
    except (OSError, ValueError):
        try:
            q = float(open('/sys/fs/cgroup/cpu/cpu.cfs_quota_us').read())
            per = float(open('/sys/fs/cgroup/cpu/cpu.cfs_period_us').read())
            if q > 0 and per > 0:
                quota = q / per
        except (OSError, ValueError):
            pass
    if quota is not None:
        cores = min(cores, max(1, int(quota + 0.5)))
    return cores


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=100)
    ap.add_argument('--warmup', type=int, default=10)
    ap.add_argument('--pre-roll', type=int, default=PRE_ROLL, help='untimed iterations before the first timed leg (power-state settle)')
    ap.add_argument('--points', type=int, default=10_000_000, help='data points per GPU')
    ap.add_argument('--strong', action='store_true', help='strong scaling: --points is the TOTAL over all GPUs (default: per GPU, weak)')
    ap.add_argument('--legs', choices=['all', 'main', 'configs'], default='all', help="'main': only the look-ahead leg that `value` is taken from "
                    "(profiling: the kernel trace then holds warm-up + pre-roll + exactly K timed iterations)")
    ap.add_argument('--cpu-sample', type=int, default=1_000_000, help='points of the cpu_baseline sample (0 = skip)')
    ap.add_argument('--min-timed', type=float, default=MIN_TIMED_S, help='repeat the main leg until this many seconds are timed (0: one repeat)')
    ap.add_argument('--multi', choices=['auto', 'on', 'off'], default='auto', help="the self-check legs of an N > 1 line (multi_gpu_parity, allreduce_us, "
                    "strong_leg, host_sum_ms_per_step): 'auto' = whenever more than one rank runs; 'on' also with one rank behind a communicator (rehearsal)")
    ap.add_argument('--only-config', type=int, default=None, help="with --legs configs: 2, 3 or 4 (one configuration per kernel trace)")
    ap.add_argument('--dry', action='store_true', help='print the schema of the multi-GPU line from a device group of compile-only members (no GPU needed)')
    args = ap.parse_args()
    if args.dry:
        return dry_line(args)

    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    # `python bench.py --gpus N` without a launcher: ONE process drives N GPUs through the library's device group
    # (gfh_create_group: one member context + host thread per GPU; the members' sums travel by RCCL all-reduce,
    # ncclCommInitAll).  Under torch.distributed.run (what the scaling runs use) it is one process per GPU with RCCL
    # (ncclCommInitRank).  Either way the line reports the ranks RCCL itself counts (`rccl_nranks`).
    group = 0
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            group = args.gpus
        else:
            args.gpus = world

    # torch first: its bundled HIP runtime must be the one the process loads
    import torch
    import torch.distributed as dist
    import numpy as np
    from gadfit_amd import _lib
    from gadfit_amd.ad import trace_model
    from tests import models as M

    # (a launcher that shows every rank only its own card -- HIP_VISIBLE_DEVICES / ROCR_VISIBLE_DEVICES per rank -- leaves one
    # visible device with index 0 whatever LOCAL_RANK says)
    if torch.cuda.device_count() and local_rank >= torch.cuda.device_count():
        local_rank = local_rank % torch.cuda.device_count()
    torch.cuda.set_device(local_rank)
    if args.legs == 'configs':
        print(json.dumps({'configs': configs_leg(_lib, M, trace_model, args.only_config)}))
        return
    # the distributed path (process group, RCCL communicator inside the library, all-reduces) is
    # also taken at world size 1 when launched through torch.distributed.run, so it can be exercised on one GPU
    use_dist = world > 1 or 'TORCHELASTIC_RUN_ID' in os.environ or os.environ.get('GADFIT_BENCH_FORCE_DIST') == '1'
    g_cpu = None
    if use_dist:
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        os.environ.setdefault('MASTER_PORT', '29517')
        dist.init_process_group('nccl', rank=rank, world_size=world, device_id=torch.device('cuda', local_rank))
        g_cpu = dist.new_group(backend='gloo')        # host-side exchanges of the self-check legs; waits that must not occupy a card

    ctx = _lib.Context(devices=group) if group else _lib.Context(local_rank)
    if group:
        world = group                       # points, `value` and n_gpus count the members; this process is "rank 0" of the report
    if use_dist:
        uid = [_lib.Context.unique_id() if rank == 0 else None]
        dist.broadcast_object_list(uid, src=0)
        ctx.comm_init(world, rank, uid[0])
    rccl_nranks = ctx.comm_info()[0]
    host_sum = os.environ.get('GADFIT_HIP_GROUP_REDUCE') == 'host'
    if world > 1 and rccl_nranks != world and not (group and host_sum):
        raise RuntimeError('%d ranks but the RCCL communicator counts %d: refusing to time a path that does not sum through RCCL'
                           % (world, rccl_nranks))

    # ---- synthetic data: this rank's slice of the global ascending-x array
    n_total = args.points if args.strong else args.points * world
    begin, count = _lib.partition(n_total, world, rank)
    truth = M.gauss8_truth()
    tape = trace_model(M.model_gauss8, 32)
    ctx.set_model(tape)
    if group:
        # the group splits the whole array itself (gfh_partition per member); `count` stays member 0's share for the roofline line.
        # Every member's slice is generated on a thread of its own (numpy releases the GIL; the generator is counter-based, so the
        # slices are the pieces of one global array) straight into the whole arrays the library cuts from: 8e7 points in one
        # numpy pass were most of a scaling run's start-up.
        from concurrent.futures import ThreadPoolExecutor
        x = np.empty(n_total); y = np.empty(n_total); sigma = np.empty(n_total)

        def fill(r):
            b, c_ = _lib.partition(n_total, world, r)
            x[b:b + c_], y[b:b + c_], sigma[b:b + c_] = M.make_single_slice(M.gauss8_numpy, truth, n_total, b, c_, 0.0, 100.0)
        with ThreadPoolExecutor(max_workers=world) as ex:
            list(ex.map(fill, range(world)))
        ctx.set_data(x, y, sigma, [0, n_total])
    else:
        x, y, sigma = M.make_single_slice(M.gauss8_numpy, truth, n_total, begin, count, 0.0, 100.0)
        ctx.set_data_local(n_total, [0, n_total], begin, x, y, sigma)
    ctx.init_weights(4)                      # USER: w = 1/sigma on the device (gadfit.F90:463-465)
    active = list(range(32)); is_global = [0] * 32
    jac, dim = ctx.jacobian_indices(active, is_global)
    start = M.start_values(truth).reshape(1, 32).copy()
    last = {}

    def steps(k, on=None, strict=True, **extra):
        """k LM iterations as fits of FIT_ITERS iterations (the last one shorter), each from `start`"""
        done = 0
        while done < k:
            n = min(FIT_ITERS, k - done)
            _, r = (on or ctx).fit(start, active, is_global, lambda_=1.0, max_iter=n, **extra)
            if r.iterations != n and (strict or r.iterations < 1):
                raise RuntimeError('fit stopped after %d of %d iterations (exit %d)' % (r.iterations, n, r.exit_reason))
            # (not strict -- the legs on other data than the headline's: a fit that ends a trial early, at a rejected step, counts its own iterations)
            last['iterations'] = last.get('iterations', 0) + r.iterations
            done += n
            last['r'] = r
            for key in ('n_sweeps', 'n_chi2', 'n_lookahead'):
                last[key] = last.get(key, 0) + getattr(r, key)

    def fence():
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()

    def timed(k, on=None, collective=True, strict=True, **extra):
        c_ = on or ctx
        last.clear()
        c_.reset_timers()
        if collective:
            fence()
        else:
            torch.cuda.synchronize()
        t0 = time.perf_counter()
        steps(k, on=c_, strict=strict, **extra)
        if collective:
            fence()
        else:
            c_.sync()
        dt = time.perf_counter() - t0
        if use_dist and collective:
            tt = torch.tensor([dt], dtype=torch.float64, device='cuda')
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            dt = float(tt.item())
        return dt, c_.timers(), dict(last), c_.timer_spread()

    def copy_rate(n_copies):
        """plain device-to-device copy of 1 GiB (read + write bytes / time): the state of the memory side on this box, now"""
        try:
            ca = torch.empty(1 << 27, dtype=torch.float64, device='cuda'); cb = torch.empty_like(ca)
            for _ in range(max(3, n_copies // 2)):
                cb.copy_(ca)
            e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(n_copies):
                cb.copy_(ca)
            e1.record(); torch.cuda.synchronize()
            return 2.0 * ca.numel() * 8 * n_copies / (e0.elapsed_time(e1) * 1e-3) / 1e9
        except Exception:
            return None

    ctx.set_lookahead(True)
    # a long job by construction: the Jacobian buffer is placed at its first sweep, inside the warm-up, rather than after the
    # library's default of 48 sweeps (which could fall into a timed leg for some --warmup / --pre-roll)
    ctx.set_placement_after(0)
    steps(max(1, args.warmup))              # kernel load + W untimed iterations
    cold = None
    if args.pre_roll > 0:
        # power-state settle (see PRE_ROLL).  Its first 20 iterations are timed too and reported as `cold_start`:
        # what a short fit right after an idle gap sees (the transient), beside the steady state of `value`.
        n_cold = min(20, args.pre_roll)
        time.sleep(0.5)
        dt_c, tm_c, _, sp_c = timed(n_cold)
        cold = {'steps': n_cold, 'ms_per_step': 1e3 * dt_c / n_cold, 'sweep_gram_avg_ms': 1e3 * tm_c[0] / max(1.0, tm_c[6]),
                'sweep_gram_max_ms': 1e3 * sp_c[1],
                'note': 'the first iterations after a 0.5 s idle gap (part of the untimed pre-roll): power-management transient'}
        if args.pre_roll > n_cold:
            steps(args.pre_roll - n_cold)
    # how fast the fused kernel's store stream runs depends on the card and on the pages behind the Jacobian buffer (DESIGN.md
    # section 3, placement); a plain
    # device-to-device copy right before the timed leg is reported beside the roofline line as a second reference
    copy_before = copy_rate(20) if rank == 0 else None
    # main leg: K timed iterations per repeat; `value` is the median repeat
    reps = []
    total = 0.0
    while True:
        reps.append(timed(args.steps))
        total += reps[-1][0]
        if total >= args.min_timed or len(reps) >= 400:
            break
    order = sorted(range(len(reps)), key=lambda k: reps[k][0])
    dt, tm, counts, spread = reps[order[len(order) // 2]]
    rep_ms = [1e3 * r[0] / args.steps for r in reps]
    n_allreduce_main = ctx.comm_info()[1]
    state_chi2 = counts['r'].chi2
    extra = args.legs == 'all'
    dt_ref = dt_nj = dt_acc = float('nan'); tm_ref = tm_nj = tm_detail = tm; counts_ref = counts_nj = counts_acc = counts; ctx_source_sha1 = None
    if extra:
        # the reference's schedule of passes, same K iterations, for comparison (not `value`)
        ctx.set_lookahead(False)
        steps(1)
        dt_ref, tm_ref, counts_ref, spread_ref = timed(args.steps)
        ctx.set_lookahead(True)
        state_chi2 = counts['r'].chi2
        # the same fits without the Jacobian store (gfh_set_keep_jacobian mode 2: a plain fit never reads J back)
        ctx.set_keep_jacobian(2)
        steps(2)
        dt_nj, tm_nj, counts_nj, _ = timed(args.steps)
        import hashlib
        ctx_source_sha1 = hashlib.sha1(ctx.model_source(active).encode()).hexdigest() if not group else None     # (the no-store kernel's source)
        ctx.set_keep_jacobian(1)
        steps(1)
        # geodesic acceleration (accth = 0.9, what the reference's own tests and examples use): STEP 3 adds the
        # omega + J^T omega kernel to every iteration (gfh_k_omega_jt; the stored Jacobian is not re-read)
        steps(1, accth=0.9)
        dt_acc, tm_acc, counts_acc, _ = timed(args.steps, accth=0.9)
        # untimed leg with events around every stage (each event record costs ~5 us of stream time, so the
        # timed legs only bracket the model kernels): reduce+assemble and all-reduce device times
        ctx.set_timer_detail(2)
        ctx.reset_timers()
        steps(min(5, args.steps))
        tm_detail = ctx.timers()
        ctx.set_timer_detail(1)
        # a fit that rejects trial steps (40 %-off start, lambda0 = 1e-6, up to 9 trials per iteration): what the look-ahead
        # schedule pays where its sweep at the trial point is thrown away (0.5 ms against the 0.1 ms chi2 of the reference schedule)
        rej = {}
        start_rej = truth * (1.0 + 0.4 * np.where(np.arange(32) % 2 == 0, 1.0, -1.0))
        for la in (1, 0):
            ctx.set_lookahead(bool(la))
            ctx.fit(start_rej.reshape(1, 32), active, is_global, lambda_=1e-6, lam_incs=8, max_iter=12)
            ctx.reset_timers()
            fence()
            t0r = time.perf_counter()
            _, rr = ctx.fit(start_rej.reshape(1, 32), active, is_global, lambda_=1e-6, lam_incs=8, max_iter=12)
            fence()
            dtr = time.perf_counter() - t0r
            tmr = ctx.timers()
            rej['lookahead' if la else 'reference_schedule'] = {
                'ms_per_iteration': 1e3 * dtr / max(1, rr.iterations), 'iterations': rr.iterations, 'exit_reason': rr.exit_reason,
                'trial_chi2_values': rr.n_chi2, 'sweep_launches': int(tmr[6]), 'chi2_launches': int(tmr[7]),
                'lookahead_sweeps_thrown_away': int(tmr[6]) - rr.n_sweeps if la else 0, 'final_chi2': rr.chi2}
        ctx.set_lookahead(True)
        rej['same_result'] = rej['lookahead']['final_chi2'] == rej['reference_schedule']['final_chi2']
    # what a plain device-to-device copy reaches on this box (read + write bytes / time; SURVEY section 8d asks for the
    # measured figure beside the 8 TB/s specification): 1 GiB buffers, 100 copies after 60 untimed ones
    copy_gbs = copy_rate(100) if (rank == 0 and extra) else None
    # tm: HIP-event device times accumulated over the timed steps, this rank's stream


    # ---- N > 1: the line proves itself.  The first run on several cards is also the first execution of RCCL over
    # more than one rank, so every such line carries: (a) the all-reduced sums against the rank-ordered host sum of the ranks' own
    # partials and a 10-iteration fit against a one-rank fit of the same points; (b) the measured latency of the all-reduce itself;
    # (c) the same iterations with the device group's ordered host sum; (d) a strong-scaling leg at --points in total.
    do_multi = args.multi == 'on' or (args.multi == 'auto' and world > 1)
    mg = None
    placement_ms = ctx.placement(); placement_copy_GBps = ctx.placement_copy_GBps()      # (of the main leg's Jacobian buffer: the legs below change the data)
    if do_multi:
        packed_doubles = dim * dim + dim + 1
        path = 'host' if (group and host_sum) else 'rccl'

        def gather_np(a):
            a = np.ascontiguousarray(a, dtype=np.float64)
            if not use_dist or world == 1:
                return [a]
            t = torch.from_numpy(a.copy()); outs = [torch.empty_like(t) for _ in range(world)]
            dist.all_gather(outs, t, group=g_cpu)
            return [o.numpy() for o in outs]

        # (a1) sums
        got = ctx.sweep(start, active, jac, dim)
        n_dev = max(1, torch.cuda.device_count())
        wrap = os.environ.get('GADFIT_HIP_GROUP_WRAP', '0') not in ('', '0')
        if group:
            parts = [_rank_partial(_lib, tape, (r % n_dev) if wrap else r, world, r, n_total, 0, x, y, sigma, active, is_global, start, False)
                     for r in range(world)]
            got_all = [np.concatenate([got[0].ravel(), got[1], [got[2]]])]
        else:
            mine = _rank_partial(_lib, tape, local_rank, world, rank, n_total, begin, x, y, sigma, active, is_global, start, True)
            flat = gather_np(np.concatenate([mine[0].ravel(), mine[1], [mine[2]]]))
            parts = [(f[:dim * dim].reshape(dim, dim), f[dim * dim:dim * dim + dim], f[-1]) for f in flat]
            got_all = gather_np(np.concatenate([got[0].ravel(), got[1], [got[2]]]))
        ref = _ordered_sum(parts)
        sums_dev = _sum_deviation(got, ref)
        bitwise_ranks = all(np.array_equal(g_, got_all[0]) for g_ in got_all)
        # (b) the all-reduce alone
        lat = ctx.debug_allreduce_latency(packed_doubles, ALLREDUCE_ROUNDS)
        # (d) strong scaling: --points in TOTAL over the ranks, same model, same iterations (the main leg is weak: --points per GPU)
        strong = None
        if not args.strong:
            ns_total = args.points
            if group:
                xs_, ys_, ss_ = M.make_single_slice(M.gauss8_numpy, truth, ns_total, 0, ns_total, 0.0, 100.0)
                ctx.set_data(xs_, ys_, ss_, [0, ns_total])
            else:
                b_, c_ = _lib.partition(ns_total, world, rank)
                xs_, ys_, ss_ = M.make_single_slice(M.gauss8_numpy, truth, ns_total, b_, c_, 0.0, 100.0)
                ctx.set_data_local(ns_total, [0, ns_total], b_, xs_, ys_, ss_)
            ctx.init_weights(4)
            steps(2 * FIT_ITERS, strict=False)
            dt_s, tm_s, cnt_s, _ = timed(args.steps, strict=False)
            k_s = max(1, cnt_s['iterations'])
            strong = {'points_total': ns_total, 'iterations_timed': k_s, 'ms_per_step': 1e3 * dt_s / k_s, 'lm_iters_per_s': k_s / dt_s,
                      'value': ns_total * k_s / dt_s, 'sweep_gram_avg_ms': 1e3 * tm_s[0] / max(1.0, tm_s[6]),
                      'note': 'the same fits with --points points in TOTAL split over the ranks (gadfit.F90:977-983); K timed iterations after 2 untimed fits'}
        # (a2) fit: PARITY_FIT_POINTS points over all ranks against the same points on one rank
        nf = PARITY_FIT_POINTS
        xf, yf, sf = M.make_single_slice(M.gauss8_numpy, truth, nf, 0, nf, 0.0, 100.0)
        if group:
            ctx.set_data(xf, yf, sf, [0, nf])
        else:
            b_, c_ = _lib.partition(nf, world, rank)
            ctx.set_data_local(nf, [0, nf], b_, xf[b_:b_ + c_], yf[b_:b_ + c_], sf[b_:b_ + c_])
        ctx.init_weights(4)
        c1 = None
        if rank == 0:
            c1 = _lib.Context(local_rank)
            c1.set_model(tape); c1.set_data(xf, yf, sf, [0, nf]); c1.init_weights(4)

        def fit_pair(act):
            """The N-rank fit against the one-rank fit, parameter by parameter.  A different number of ranks is a different order of
            additions (as co_sum over another number of images is in the reference), and LM hands such last-bit differences of J^T J /
            J^T r to the parameters through the inverse of the normal matrix: a parameter the data barely determine (this workload's
            skews: 0.01-0.03 with standard errors of their own size, correlated with the centres) moves by far more than 1e-10 of its
            VALUE while chi2 agrees to the last bits.  So a deviation counts against the bound 1e-10 |p| + 1e-9 sigma_p (sigma_p from
            the inverse normal matrix at the solution: a billionth of the parameter's own standard error), and the same pair of fits
            with the skews held fixed must meet the plain 1e-10 |p|."""
            pN, rN = ctx.fit(start, act, is_global, lambda_=1.0, max_iter=PARITY_FIT_ITERS)
            pN_all = gather_np(pN.ravel())
            if rank != 0:
                return None
            p1, r1 = c1.fit(start, act, is_global, lambda_=1.0, max_iter=PARITY_FIT_ITERS)
            j1, d1 = c1.jacobian_indices(act, is_global)
            H, _, chi_at = c1.sweep(p1, act, j1, d1)
            sig = np.zeros(32)
            sig[np.asarray(act)] = np.sqrt(np.abs(np.diag(np.linalg.inv(H))) * chi_at / max(1, nf - d1))
            dev = np.abs(pN - p1).ravel(); p1f = np.abs(p1).ravel()
            rel = dev / p1f
            bound = PARITY_FIT_TOL * p1f + 1e-9 * sig
            k = int(np.argmax(dev / bound))
            same_n = rN.iterations == r1.iterations
            return {'points': nf, 'active_params': len(act), 'iterations': [int(rN.iterations), int(r1.iterations)],
                    'max_rel_dev_pars': float(np.max(rel)) if same_n else float('inf'),
                    'max_dev_over_bound': float(np.max(dev / bound)) if same_n else float('inf'),
                    'max_dev_in_sigma': float(np.max(dev[np.asarray(act)] / sig[np.asarray(act)])),
                    'worst_parameter': {'index': k, 'value': float(p1.ravel()[k]), 'sigma': float(sig[k]), 'abs_dev': float(dev[k])},
                    'rel_dev_chi2': float(abs(rN.chi2 - r1.chi2) / r1.chi2), 'tol': PARITY_FIT_TOL,
                    'ranks_agree_bitwise': bool(all(np.array_equal(q, pN_all[0]) for q in pN_all))}
        fit = fit_pair(active)
        fit24 = fit_pair([k for k in range(32) if k % 4 != 3])
        if rank == 0:
            c1.close()
            fit['what'] = ('%d-iteration fit of %d points split over the %d ranks against the same fit on one rank (fresh context, no communicator); bound per '
                           'parameter: 1e-10 |p| + 1e-9 sigma_p (this workload\'s skews are barely determined: see skews_fixed for the plain 1e-10)' % (PARITY_FIT_ITERS, nf, world))
            fit24['what'] = 'the same pair of fits with the 8 skew parameters held at their start values (24 active): plain relative bound 1e-10'
            fit['skews_fixed'] = fit24
            fit['ok'] = bool(fit['max_dev_over_bound'] <= 1.0 and fit['ranks_agree_bitwise'] and fit24['max_rel_dev_pars'] <= PARITY_FIT_TOL and fit24['ranks_agree_bitwise'])
        mg = (path, sums_dev, bitwise_ranks, fit, lat, packed_doubles, strong)

    out = None
    if rank == 0:
        n_sweep = max(1.0, tm[6]); n_chi2 = max(1.0, tm[7])
        # the dominant kernel's average over the WHOLE timed region (all repeats): the library brackets every 8th launch with
        # HIP events and scales the sum to the launches, so sum(tm[0]) / sum(tm[6]) is the mean over the timed launches
        sweep_ms = 1e3 * sum(r[1][0] for r in reps) / max(1.0, sum(r[1][6] for r in reps))
        launches_timed = int(sum(r[3][3] for r in reps))
        spread = (min(r[3][0] for r in reps if r[3][3] > 0), max(r[3][1] for r in reps), spread[2], launches_timed)
        gram_ms = 1e3 * tm[1] / n_sweep
        chi2_ms = 1e3 * tm[4] / n_chi2
        achieved = SWEEP_BYTES_PER_POINT * count / (sweep_ms * 1e-3) / 1e9
        fused = os.environ.get('GADFIT_HIP_FUSED', '1') != '0'
        kernel_name = ('gfh_k_sweep_gram (residual + Jacobian AD sweep fused with J^T J / J^T r on FP64 MFMA; J written once)'
                       if fused else 'gfh_k_sweep (residual + Jacobian AD sweep)')
        # HBM bytes per launch from the COMMITTED rocprofv3 PMC passes of this same command (FETCH_SIZE x2 gfx950 correction
        # + WRITE_SIZE, profiles/traffic.json) -- not collected in this run; only valid for the profiled size
        traffic = None
        mfma_util = None
        try:
            tj = json.load(open(os.path.join(ROOT, 'profiles', 'traffic.json')))
            if tj.get('points') == count and world == 1:
                kn = 'gfh_k_sweep_gram' if fused else 'gfh_k_sweep'
                traffic = tj['hbm_bytes_per_launch'].get(kn)
                mfma_util = tj.get('mfma', {}).get('gfh_k_sweep_gram' if fused else 'gfh::k_gram<2>', {}).get('util')
        except Exception:
            pass
        out = {
            'metric': 'LM iterations/s x data points, N=1e7 pts/GPU x 32 active params (whole job)',
            'value': n_total * args.steps / dt,
            'unit': 'point-iterations/s',
            'lm_iters_per_s': args.steps / dt,
            'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup,
            'ms_per_step': 1e3 * dt / args.steps,
            'repeats': {'n': len(reps), 'timed_s': total, 'ms_per_step_median': 1e3 * dt / args.steps, 'ms_per_step_min': min(rep_ms),
                        'ms_per_step_max': max(rep_ms), 'note': 'K timed iterations per repeat; value and ms_per_step are the median repeat'},
            'rccl_nranks': rccl_nranks,
            'cross_rank_sum': ('ordered host sum of the device group (GADFIT_HIP_GROUP_REDUCE=host)' if (group and host_sum) else
                               'ncclAllReduce of the packed [JTJ|JTres|chi2|status] (%d doubles) per sweep, of [chi2|status] per chi2()'
                               % (dim * dim + dim + 2)) if world > 1 else 'none (one image)',
            'allreduces_in_main_leg': n_allreduce_main,
            'higher_is_better': True, 'scaling': 'strong' if args.strong else 'weak', 'vs_baseline': None,
            'dtype': 'f64', 'data': 'synthetic',
            'config': {'workload': 'gauss8: 8 skewed Gaussians, 32 active params, %d pts/GPU, sigma given (USER), '
                                   'gfh_fit: fits of %d LM iterations from 5%%-off start values, lambda0=1, lambda x/÷10, '
                                   'look-ahead schedule' % (count, FIT_ITERS),
                       'points_total': n_total, 'active_params': 32, 'partition': 'contiguous, gadfit.F90:977-983',
                       'parallelism': ('single-process device group: %d member threads, %s' % (group, 'ordered host sum' if host_sum else 'RCCL all-reduce (ncclCommInitAll)')) if group else
                                      ('one process per GPU, RCCL all-reduce' if world > 1 else 'one GPU'),
                       'pre_roll_steps': args.pre_roll,
                       'pre_roll': 'untimed iterations before the first timed leg, on top of --warmup: after an idle gap the '
                                   'part slows launches ~3-40 of a back-to-back series (tools/probes/transient.py); the timed legs are steady state'},
            'roofline': {'bound': 'hbm', 'kernel': kernel_name,
                         'achieved': achieved, 'peak': HBM_PEAK_GBS, 'unit': 'GB/s', 'frac': achieved / HBM_PEAK_GBS,
                         'traffic': traffic,
                         'traffic_source': None if traffic is None else 'profiles/traffic.json: committed rocprofv3 --pmc passes of this command '
                                           '(tools/pmc_fused.sh), per launch; a committed constant, not measured in this run',
                         'bytes_per_point': SWEEP_BYTES_PER_POINT, 'points_per_launch': count,
                         'avg_ms': sweep_ms,
                         'copy_GBps_measured_on_this_box': copy_gbs,   # torch device-to-device copy, read + write bytes (after all legs)
                         'copy_GBps_before_main_leg': copy_before,     # the same right before the timed leg: the memory side's state then
                         'frac_of_measured_copy': (achieved / copy_gbs) if copy_gbs else None,
                         # the same kernel's shortest and longest launch in the timed region (launch-to-launch spread,
                         # DESIGN.md section 3); `achieved` is the average over the timed region, not the best launch
                         # the library brackets every 8th launch with HIP events (two event records per launch cost 8 us of stream
                         # time per iteration); avg_ms / min_ms / max_ms are over the launches that were timed in all repeats
                         'launches_timed': int(spread[3]), 'launches_in_timed_region': int(sum(r[1][6] for r in reps)),
                         # the Jacobian buffer's placement (gfh_set_placement_tries): the kernel's ms on the allocation kept, then
                         # on the candidates that were freed -- the physical pages behind the buffer decide 0.46 ... 0.52 ms
                         'jacobian_placement_ms': placement_ms,
                         # the device-to-device copy rate the library measured inside the first candidate and scaled its "fast side" thresholds with
                         'jacobian_placement_copy_GBps': placement_copy_GBps,
                         'min_ms': 1e3 * spread[0], 'max_ms': 1e3 * spread[1],
                         'frac_best_launch': (SWEEP_BYTES_PER_POINT * count / max(spread[0], 1e-12) / 1e9) / HBM_PEAK_GBS},
            'kernels_ms': {'sweep': sweep_ms, 'gram_mfma': 1e3 * tm_detail[1] / max(1.0, tm_detail[6]),
                           'reduce_assemble': 1e3 * tm_detail[2] / max(1.0, tm_detail[6]),
                           'allreduce': 1e3 * tm_detail[3] / max(1.0, tm_detail[6]), 'chi2': chi2_ms},
            'gram': ({'fused_into_sweep': True, 'fp64_matrix_peak_TFLOPs': 78.6,
                      'mfma_busy_frac_rocprof': mfma_util,     # SQ_VALU_MFMA_BUSY_CYCLES / (SIMDs x cycles), profiles/
                      # per 4 points and wave: one 16x16x4 (off-diagonal tile, 2048 flop) + five 4x4x4_4b (the 20 distinct 4x4
                      # blocks of the two diagonal tiles, 512 flop each): 4608 flop issued for 2 x (528 + 32) useful ones x 4
                      'fp64_mfma_TFLOPs_issued': (2048 + 5 * 512) * (count / 4.0) / (sweep_ms * 1e-3) / 1e12}
                     if fused else
                     {'achieved_GBps': GRAM_BYTES_PER_POINT * count / (gram_ms * 1e-3) / 1e9,
                      'fp64_mfma_TFLOPs_issued': 3 * 2048 * (count / 4.0) / (gram_ms * 1e-3) / 1e12}),
            'chi2_GBps': CHI2_BYTES_PER_POINT * count / (chi2_ms * 1e-3) / 1e9,
            'passes_in_timed_region': {'iterations': args.steps, 'sweep_gram_launches': int(tm[6]), 'chi2_launches': int(tm[7]),
                                       'trial_chi2_from_lookahead_sweep': counts['n_lookahead']},
            'reference_schedule': None if not extra else {'ms_per_step': 1e3 * dt_ref / args.steps, 'lm_iters_per_s': args.steps / dt_ref,
                                   'value': n_total * args.steps / dt_ref,
                                   'sweep_gram_launches': int(tm_ref[6]), 'chi2_launches': int(tm_ref[7]),
                                   'chi2_ms': 1e3 * tm_ref[4] / max(1.0, tm_ref[7])},
            'jacobian_not_kept': None if not extra else {'kernel': 'gfh_k_sweep_gram_nostore', 'ms_per_step': 1e3 * dt_nj / args.steps, 'lm_iters_per_s': args.steps / dt_nj,
                                  'sweep_gram_ms': 1e3 * tm_nj[0] / max(1.0, tm_nj[6]),
                                  # its bound is the FP64 pipe of a SIMD, which vector and matrix instructions share: per 64-point pass of a
                                  # wave 392 FP64 VALU instructions + 16 x (one 16x16x4 + five 4x4x4_4b matrix instructions) = 1630 ns of
                                  # pipe time whatever the number of resident waves (tools/microbench/fp64_phases.hip, profiles/r04_nostore.md:
                                  # the pipe alone, no LDS, no memory), N / 64 / 1024 such passes per SIMD
                                  'roofline': nostore_roofline(ctx_source_sha1, count, 1e3 * tm_nj[0] / max(1.0, tm_nj[6])),
                                  'note': 'gfh_set_keep_jacobian(2): same fits, the fused kernel skips the 8*p B/point Jacobian store '
                                          '(nothing in a plain fit reads J back; the mode the Fortran / Python gadf_fit layers ask for); FP64-pipe-bound, not part of `value`',
                                  'same_result': bool(counts_nj['r'].chi2 == counts['r'].chi2)},
            'accelerated_fit': None if not extra else {'accth': 0.9, 'ms_per_step': 1e3 * dt_acc / args.steps, 'lm_iters_per_s': args.steps / dt_acc,
                                'omega_passes': counts_acc['r'].n_omega, 'note': 'same fits with geodesic acceleration (STEP 3, gadfit.F90:715-743): '
                                'one gfh_k_omega_jt launch per iteration on top of the fused sweep; not part of `value`'},
            'rejecting_fit': None if not extra else rej,
            'cold_start': cold,
            'final_chi2_per_dof': state_chi2 / (n_total - dim),
        }
    ctx.close()

    # (c) the same iterations with the device group's ordered host sum in place of RCCL: one process, a member thread per card
    # (gfh_create_group under GADFIT_HIP_GROUP_REDUCE=host).  Under torch.distributed.run rank 0 alone drives all cards for this leg,
    # after every rank has closed its context; the others wait on the host (gloo), not on their cards.
    if do_multi:
        host_leg = None
        if group and host_sum:
            host_leg = {'ms_per_step': 1e3 * dt / args.steps, 'note': 'the main leg itself ran on the ordered host sum (GADFIT_HIP_GROUP_REDUCE=host)'}
        else:
            if use_dist:
                dist.barrier(group=g_cpu)
            if rank == 0:
                keep = os.environ.get('GADFIT_HIP_GROUP_REDUCE')
                os.environ['GADFIT_HIP_GROUP_REDUCE'] = 'host'
                gh = None
                try:
                    if world > max(1, torch.cuda.device_count()) and os.environ.get('GADFIT_HIP_GROUP_WRAP', '0') in ('', '0'):
                        raise RuntimeError('%d members but %d visible cards' % (world, torch.cuda.device_count()))
                    if not group:
                        from concurrent.futures import ThreadPoolExecutor
                        xw = np.empty(n_total); yw = np.empty(n_total); sw = np.empty(n_total)

                        def fill_h(r):
                            b, c_ = _lib.partition(n_total, world, r)
                            xw[b:b + c_], yw[b:b + c_], sw[b:b + c_] = M.make_single_slice(M.gauss8_numpy, truth, n_total, b, c_, 0.0, 100.0)
                        with ThreadPoolExecutor(max_workers=world) as ex:
                            list(ex.map(fill_h, range(world)))
                    else:
                        xw, yw, sw = x, y, sigma
                    gh = _lib.Context(devices=world)
                    gh.set_model(tape); gh.set_data(xw, yw, sw, [0, n_total]); gh.init_weights(4)
                    gh.set_lookahead(True); gh.set_placement_after(0)
                    steps(max(FIT_ITERS, min(args.pre_roll, 40)), on=gh)
                    dt_h, tm_h, cnt_h, _ = timed(args.steps, on=gh, collective=False)
                    host_leg = {'ms_per_step': 1e3 * dt_h / args.steps, 'lm_iters_per_s': args.steps / dt_h, 'value': n_total * args.steps / dt_h,
                                'sweep_gram_avg_ms': 1e3 * tm_h[0] / max(1.0, tm_h[6]), 'rccl_nranks': gh.comm_info()[0],
                                'host_sum_us': gh.debug_allreduce_latency(dim * dim + dim + 1, ALLREDUCE_ROUNDS),
                                'same_result_as_rccl': bool(cnt_h['r'].chi2 == counts['r'].chi2) if not args.strong else None,
                                'note': 'one process, %d member threads, sums added on the host in rank order from the members\' pinned mailboxes '
                                        '(bitwise the same on every member); same points, same fits, K timed iterations after an untimed pre-roll' % world}
                except Exception as e:              # an auxiliary leg: the line is not lost over it, but says so
                    host_leg = {'ms_per_step': None, 'error': repr(e)}
                finally:
                    if gh is not None:
                        gh.close()
                    if keep is None:
                        os.environ.pop('GADFIT_HIP_GROUP_REDUCE', None)
                    else:
                        os.environ['GADFIT_HIP_GROUP_REDUCE'] = keep
            if use_dist:
                dist.barrier(group=g_cpu)
        if rank == 0:
            out.update(_multi_gpu_block(world, mg[0], mg[1], mg[2], mg[3], mg[4], mg[5], mg[6], host_leg,
                                        None if mg[0] != 'rccl' else 1e3 * dt / args.steps))

    # ---- what a user pays before the iterations (never `value`): every step from a FRESH context to the end of a first
    # 10-iteration fit at this size, and the same through the Fortran API (tests/fortran/bench_headline.F90, where recording
    # eval() over the data is part of the first gadf_fit) when that program is built.
    if rank == 0 and world == 1 and extra:
        try:
            out['setup'] = setup_leg(_lib, M, trace_model, truth, x, y, sigma, count, active, is_global, start)
        except Exception as e:          # the line must not be lost over an auxiliary leg
            out['setup'] = {'error': repr(e)}
        # what a one-fit user of the Fortran API waits for, at top level: `value` is a steady state behind the pre-roll, this is not
        fa = out['setup'].get('fortran_api') or {}
        out['first_gadf_fit_ms'] = fa.get('first_gadf_fit_ms')
        out['gadf_init_to_end_of_first_gadf_fit_ms'] = (None if fa.get('first_gadf_fit_ms') is None else
                                                        fa.get('gadf_init_add_dataset_set_ms', 0.0) + fa['first_gadf_fit_ms'])
        out['first_gadf_fit_note'] = ('tests/fortran/bench_headline.F90 under the DEFAULT capture (eval() recorded at every abscissa), %d iterations; '
                                      'phases in setup.fortran_api' % FIT_ITERS)

    # ---- BASELINE.json configs 2-4 on this card, in the same line (never `value`)
    if rank == 0 and world == 1 and extra:
        try:
            out['configs'] = configs_leg(_lib, M, trace_model)
        except Exception as e:
            out['configs'] = {'error': repr(e)}

    # ---- CPU baseline.  Since round 6 the timed CPU path is the REFERENCE'S OWN C++ AD and linear algebra (oracle/_ref/
    # libgadfit_refcxx.so: automatic_differentiation.cpp, fit_function.cpp, lapack_fallback.cpp compiled from where they lie, under
    # oracle/ref_cxx_driver.cpp, whose per-point loop is lm_solver.cpp:286-346, 513-529) -- kind "reference" -- wherever that file
    # travelled with the snapshot; the oracle (C restatement of the Fortran side, kind "port") is timed beside it on the same
    # sample, and the build container's calibration of the two (tools/calibrate_cpu_baseline.py) is read from the committed file.
    if rank == 0 and world == 1 and args.cpu_sample > 0:      # N = 1 only: the scaling runs do not re-time the host
        from oracle import binding as orc
        from oracle import refcxx
        ns = args.cpu_sample
        xs, ys, ss = M.make_single_slice(M.gauss8_numpy, truth, ns, 0, ns, 0.0, 100.0)
        p = orc.OracleProblem(tape, [xs], [ys], [1.0 / ss], [M.start_values(truth)], active, is_global)
        iters = 2
        c0 = time.perf_counter()
        for _ in range(iters):
            p.sweep(); p.chi2()
        cdt = time.perf_counter() - c0
        port = {'value': ns * iters / cdt, 'unit': 'point-iterations/s', 'cores': 1, 'kind': 'port',
                'sample': '%d points x %d iterations (sweep + chi2) of the same gauss8 workload, '
                          'oracle/gadfit_oracle.c single thread' % (ns, iters),
                'ns_per_point_iteration': 1e9 * cdt / (ns * iters)}
        calib = None
        try:
            cj = json.load(open(os.path.join(ROOT, 'profiles', 'r06_cpu_calibration.json')))
            g8 = next(m for m in cj['models'] if m['model'] == 'gauss8')
            calib = {'model': 'gauss8', 'points': g8['points'], 'reference_ns_1t': g8['reference_ns_1t'], 'reference_ns_8t': g8['reference_ns_8t'],
                     'port_ns_1t': g8['port_ns_1t'], 'ratio': g8['ratio'], 'host': cj['host']['cpu_model'],
                     'source': 'profiles/r06_cpu_calibration.json (tools/calibrate_cpu_baseline.py, build container: reference C++ AD + vendored '
                               'dsyrk/dgemv against the oracle on identical inputs; ratio = port / reference at 1 thread)'}
        except Exception:
            pass
        if refcxx.available():
            start_ref = M.start_values(truth)
            refcxx.sweep(refcxx.GAUSS8, xs[:20000], ys[:20000], ss[:20000], start_ref, want_J=False)
            c0 = time.perf_counter()
            for _ in range(iters):
                refcxx.sweep(refcxx.GAUSS8, xs, ys, ss, start_ref, threads=1, want_J=False)
                refcxx.chi2(refcxx.GAUSS8, xs, ys, ss, start_ref, threads=1)
            rdt = time.perf_counter() - c0
            out['cpu_baseline'] = {'value': ns * iters / rdt, 'unit': 'point-iterations/s', 'cores': 1, 'kind': 'reference',
                                   'sample': '%d points x %d iterations (STEP 1 + STEP 2 + one chi2) of the same gauss8 workload through the reference\'s '
                                             'own C++ AD (gadfit::AdVar, returnSweep) and vendored dsyrk/dgemv, 1 OpenMP thread; loop = '
                                             'lm_solver.cpp:286-346, 513-529 (oracle/ref_cxx_driver.cpp)' % (ns, iters),
                                   'ns_per_point_iteration': 1e9 * rdt / (ns * iters),
                                   'port_on_the_same_sample': port, 'port_over_reference_here': (cdt / rdt),
                                   'calibration': calib}
            # ... and with as many OpenMP threads as this job may use (how the C++ reference parallelises: lm_solver.cpp:291)
            cores = max(1, min(_cpu_share(), 128))
            refcxx.sweep(refcxx.GAUSS8, xs[:20000], ys[:20000], ss[:20000], start_ref, threads=cores, want_J=False)
            c0 = time.perf_counter()
            for _ in range(iters):
                refcxx.sweep(refcxx.GAUSS8, xs, ys, ss, start_ref, threads=cores, want_J=False)
                refcxx.chi2(refcxx.GAUSS8, xs, ys, ss, start_ref, threads=cores)
            rdt_all = time.perf_counter() - c0
            out['cpu_baseline_reference_all_cores'] = {'value': ns * iters / rdt_all, 'unit': 'point-iterations/s', 'cores': cores, 'kind': 'reference',
                                                       'sample': 'the same sample and passes on %d OpenMP threads (its fallback dsyrk scales poorly, BASELINE.md section 2)' % cores,
                                                       'ns_per_point_iteration': 1e9 * rdt_all / (ns * iters)}
        else:
            port['calibration'] = calib
            port['note'] = 'oracle/_ref/libgadfit_refcxx.so did not travel with this snapshot: the port alone, with the build container\'s calibration'
            out['cpu_baseline'] = port
    # all host cores: one process per core, each an "image" with its contiguous share of a bounded
    # sample (the reference's own parallel model, gadfit.F90:977-1002); started together, timed to the last finisher
    if rank == 0 and world == 1 and args.cpu_sample > 0:
        import subprocess
        cores = max(1, min(_cpu_share(), 128))
        per = max(50_000, args.cpu_sample // 4)
        iters = 3
        start_epoch = time.time() + 8.0 + 0.05 * cores     # interpreter + numpy start-up of every worker
        procs = [subprocess.Popen([sys.executable, os.path.join(ROOT, 'oracle', 'bench_worker.py'), str(per * cores), str(i * per),
                                   str(per), str(iters), repr(start_epoch)], stdout=subprocess.PIPE, text=True,
                                  env=dict(os.environ, OMP_NUM_THREADS='1', OPENBLAS_NUM_THREADS='1'))
                 for i in range(cores)]
        recs = []
        for pr in procs:
            o, _ = pr.communicate(timeout=600)
            if pr.returncode == 0 and o.strip():
                recs.append(json.loads(o.strip().splitlines()[-1]))
        if len(recs) == cores and all(r['t0'] - start_epoch < 0.5 for r in recs):
            wall = max(r['t1'] for r in recs) - min(r['t0'] for r in recs)
            out['cpu_baseline_all_cores'] = {'value': per * cores * iters / wall, 'unit': 'point-iterations/s', 'cores': cores,
                                             'kind': 'port', 'sample': '%d processes x %d points x %d iterations (sweep + chi2), '
                                             'one oracle process per core on contiguous shares, as coarray images' % (cores, per, iters),
                                             'wall_s': wall}
        else:
            out['cpu_baseline_all_cores'] = {'value': None, 'note': 'workers did not start together (%d of %d reported)' % (len(recs), cores)}
    if rank == 0:
        print(json.dumps(out))
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()
    # a line whose own cross-rank check failed is printed (the record exists) and the run fails
    if rank == 0 and do_multi and not out['multi_gpu_parity']['ok']:
        sys.stdout.flush()
        sys.exit('multi_gpu_parity failed: ' + json.dumps(out['multi_gpu_parity']))


if __name__ == '__main__':
    main()
