"""The auxiliary legs of bench.py (never `value`): BASELINE configs 2-4 with their rooflines, the no-store kernel's roofline, what a user
pays before the iterations (`setup`).  Split off bench.py in round 6 (it had grown past a thousand lines); bench.py imports these.
The oracle is NOT used here: config 4's algorithmic floor comes from the device's own mesh records."""
import json
import os
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HBM_PEAK_GBS = 8000.0          # MI355X HBM3E peak (MI355X_MICROARCH.md: 8.0 TB/s spec; ~6.3 achievable)
P_ACTIVE = 32
FIT_ITERS = 10                 # LM iterations per gfh_fit call in the timed region (bench.py)
NOSTORE_FLOP_PER_POINT = 2 * (P_ACTIVE * (P_ACTIVE + 1) // 2 + P_ACTIVE + 1) + 500      # Gram sums + AD body (profiles/r05_fused_kernel.md section 2)

FP64_WAVE_INSTR_PER_S = 1024 * 2.4e9 / 4.0      # a wave-level FP64 VALU instruction holds its SIMD for 4 cycles (profiles/r03_cfg4.md)
CFG4_PMC_FILE = 'profiles/r06_cfg4_pmc.json'    # tools/pmc_cfg4_r06.sh: SQ_ACTIVE_INST_VALU per launch of the CURRENT kernels, keyed by the sha1 of their source

# Necessary arithmetic of ONE integrand evaluation, in FP64 pipe instructions of the peak's kind (one fused multiply-add per lane =
# 2 flop; 78.6 TFLOP/s = 1024 SIMDs x 64 lanes x 2 flop x 2.4 GHz / 4 cycles): add / subtract / multiply / divide 1; exp 14 and log 16
# (range reduction + a degree-11/12 polynomial + reconstruction: the least a 1-ulp fp64 implementation spends); a**b = exp(b log a) 31;
# a**n by squaring; trigonometric / hyperbolic / inverse functions 20; erf 24; sqrt 4.  The GRADIENT costs on top: per operation the
# multiply-adds of its adjoint rule.  A convention, stated here so that the fraction can be recomputed; the
# kernels' own instruction counts (device-library exp / log / pow at full accuracy, interval search, error sums) are the other floor.
def integrand_instr(tape, sub):
    """(value, gradient extra) necessary FP64 instructions of one evaluation of sub-tape `sub` (include/gadfit_tape.h)"""
    from gadfit_amd import tape as T
    val = {T.ADD: 1, T.SUB: 1, T.MUL: 1, T.DIV: 1, T.POW: 31, T.EXP: 14, T.LOG: 16, T.SQRT: 4, T.ABS: 0, T.ERF: 24}
    grad = {T.ADD: 2, T.SUB: 2, T.MUL: 2, T.DIV: 3, T.POW: 5, T.EXP: 1, T.LOG: 2, T.SQRT: 2, T.ABS: 1, T.ERF: 16}
    nodes, _ = tape.subtapes[sub]
    v = g = 0
    for op, a, b, fl, c in nodes:
        if op == T.POWI:
            k = max(1, abs(int(b)).bit_length() + bin(abs(int(b))).count('1') - 2); v += k; g += 2 + k
        elif op in val:
            v += val[op]; g += 0 if (fl & T.F_REAL) else grad[op]
        elif T.SIN <= op <= T.ATANH:
            v += 20; g += 0 if (fl & T.F_REAL) else 21
    return v, g


NOSTORE_PMC_FILE = 'profiles/r06_nostore_pmc.json'     # tools/pmc_nostore_r06.sh: instruction counts per launch of the CURRENT kernel


def nostore_roofline(sha, count, kernel_ms):
    """gfh_k_sweep_gram_nostore against the FP64 pipe of a SIMD, which vector and matrix instructions share (their times add:
    tools/microbench/fp64_phases.hip, profiles/r04_nostore.md).  `frac`: the kernel's own instruction count of THIS round's counter
    pass (SQ_INSTS_VALU includes the matrix instructions; of SQ_INSTS_VALU_MFMA_F64 one in six is a 64-cycle 16x16x4, five are
    17.5-cycle 4x4x4_4b: codegen.cpp, GFH_K_SWEEP_GRAM), priced at 4 / 64 / 17.5 cycles per instruction and SIMD, refused when the
    counts were taken on another source; `flops_frac`: the necessary arithmetic against the 78.6 TFLOP/s peak."""
    r = {'bound': 'fp64 pipe (VALU + MFMA share it)', 'necessary_flop_per_point': NOSTORE_FLOP_PER_POINT, 'fp64_peak_TFLOPs': 78.6,
         'flops_frac': NOSTORE_FLOP_PER_POINT * count / (1e-3 * kernel_ms) / 78.6e12}
    try:
        pj = json.load(open(os.path.join(ROOT, NOSTORE_PMC_FILE)))
        if pj.get('source_sha1') != sha:
            r.update(floor_ms=None, frac=None, floor_source='%s is STALE: counted on source %s, the kernel that ran is %s -- re-run tools/pmc_nostore_r06.sh'
                                                           % (NOSTORE_PMC_FILE, str(pj.get('source_sha1'))[:12], str(sha)[:12]))
            return r
        c_ = pj['counters']
        scale = count / float(pj['points'])
        mfma = c_['SQ_INSTS_VALU_MFMA_F64'] * scale
        valu = c_['SQ_INSTS_VALU'] * scale - mfma
        cycles = valu * 4.0 + mfma * (64.0 + 5 * 17.5) / 6.0
        floor = 1e3 * cycles / (1024 * 2.4e9)
        r.update(floor_ms=floor, frac=floor / kernel_ms, source_sha1=sha,
                 floor_source='%s: %.4g vector + %.4g matrix wave-level instructions per launch (rocprofv3 --pmc pass of this round on the kernel of this '
                              'sha1) x 4 / (64 + 5 x 17.5) / 6 cycles, / (1024 SIMDs x 2.4 GHz)' % (NOSTORE_PMC_FILE, valu, mfma))
    except (OSError, ValueError, KeyError) as ex:
        r.update(floor_ms=None, frac=None, floor_source='%s not readable (%r)' % (NOSTORE_PMC_FILE, ex))
    return r


def configs_leg(_lib, M, trace_model, only=None, reps=100):
    """BASELINE.json configs[1..3] (the headline is configs[4] on one card): per configuration the dominant kernel's HIP-event
    time after a pre-roll of 40 launches, its algorithmic bytes (cfg 4: its FP64 VALU-issue floor), the fraction, and the wall time of
    an LM iteration of gfh_fit.  Never part of `value`."""
    import numpy as np
    out = {}

    def one(key, name, tape, xs, ys, ws, pars, active, is_global, which, kernel, bytes_pp, fit_iters, roofline_fn=None, n_reps=reps):
        t0 = time.perf_counter()
        ctx = _lib.Context(0)
        try:
            ctx.set_placement_after(1 << 30)     # (the Jacobian buffer is placed below, once the part's clocks are up)
            pos = np.zeros(len(xs) + 1, dtype=np.int64)
            for i, a in enumerate(xs):
                pos[i + 1] = pos[i] + len(a)
            n = int(pos[-1])
            ctx.set_model(tape)
            ctx.set_data(np.concatenate(xs), np.concatenate(ys), np.concatenate(ws), pos)
            jac, dim = ctx.jacobian_indices(active, is_global)
            ctx.sweep(pars, active, jac, dim)
            ctx.time_kernel(which, max(40, n_reps))            # pre-roll: the first ~40 launches after an idle gap run in the power-management transient
            # the library's placement search (up to 16 allocations of the Jacobian buffer timed, the fastest kept: context.cpp,
            # place_jacobian_now) runs at this sweep -- behind the pre-roll, so that its candidates are compared at settled clocks
            # (a process's very first launches timed 0.173 ms where the same box gives 0.149-0.152 once warm)
            ctx.set_placement_after(0)
            ctx.sweep(pars, active, jac, dim)
            ctx.time_kernel(which, 20)
            ms = ctx.time_kernel(which, n_reps)
            e = {'workload': name, 'points': n, 'n_active': len(active), 'dim': dim, 'kernel': kernel, 'kernel_ms': ms, 'launches_timed': n_reps,
                 # (the kernel's ms on the Jacobian allocation kept, then on the candidates that were freed; copy rate the thresholds were scaled with)
                 'jacobian_placement_ms': ctx.placement(), 'jacobian_placement_copy_GBps': ctx.placement_copy_GBps()}
            if roofline_fn is None:
                gbs = bytes_pp * n / (ms * 1e-3) / 1e9
                e['roofline'] = {'bound': 'hbm', 'bytes_per_point': bytes_pp, 'achieved': gbs, 'peak': HBM_PEAK_GBS, 'unit': 'GB/s', 'frac': gbs / HBM_PEAK_GBS}
                # HBM bytes per launch from the committed counter passes of this very leg (tools/pmc_configs.sh: FETCH_SIZE x 2 +
                # WRITE_SIZE, KiB; profiles/r06_configs_traffic.json) -- not collected in this run
                try:
                    tj = json.load(open(os.path.join(ROOT, 'profiles', 'r06_configs_traffic.json'))).get(key)
                    if tj:
                        e['roofline'].update(traffic=tj['hbm_bytes_per_launch'], traffic_over_algorithmic=tj['hbm_bytes_per_launch'] / (bytes_pp * n),
                                             traffic_source='profiles/r06_configs_traffic.json: committed rocprofv3 --pmc passes of `bench.py --legs configs --only-config N`, per launch; a committed constant')
                except (OSError, ValueError, KeyError):
                    pass
            else:
                e['roofline'] = roofline_fn(ctx, ms, n)
            ctx.set_keep_jacobian(2)
            ctx.fit(pars, active, is_global, lambda_=1.0, max_iter=2)
            # fits of `fit_iters` iterations from the start values, four of them: iterations of a fit in progress (a fit left to run on
            # converges within 8-9 iterations on these workloads and then spends chi2() passes on rejected trials: those are counted below)
            t1 = time.perf_counter(); its = sweeps = chis = 0
            for _ in range(4):
                _, r = ctx.fit(pars, active, is_global, lambda_=1.0, max_iter=fit_iters)
                its += r.iterations; sweeps += r.n_sweeps; chis += r.n_chi2 - r.n_lookahead
            e['lm_iteration_ms'] = 1e3 * (time.perf_counter() - t1) / max(1, its)
            e['lm_iterations_timed'] = its
            e['passes_per_iteration'] = {'sweeps': sweeps / max(1, its), 'chi2_kernels': chis / max(1, its)}
            e['lm_iteration_note'] = ('gfh_fit, look-ahead schedule, Jacobian stored only if the options read it back (what gadf_fit asks for); '
                                      '4 fits of %d iterations from the start values' % fit_iters)
        except Exception as ex:                                    # an auxiliary leg
            e = {'workload': name, 'error': repr(ex)[:300]}
        finally:
            ctx.close()
        e['leg_seconds'] = time.perf_counter() - t0
        out[key] = e

    if only in (None, 2):
        x, y, s = M.make_single(M.exp4_numpy, M.EXP4_TRUTH, 10_000_000, 0.0, 100.0)
        one('cfg2', 'single curve, 4-exponential decay, N=1e7, 8 active params', trace_model(M.model_exp4, 8), [x], [y], [1 / s],
            M.start_values(M.EXP4_TRUTH).reshape(1, 8), list(range(8)), [0] * 8, 5, 'gfh_k_sweep_gram (8 active: per-lane outer product)', 32 + 8 * 8, 5)
    if only in (None, 3):
        xs, ys, ss, truths = M.make_global7(64, 100_000)
        pars = np.array([M.start_values(t) for t in truths]); pars[:, 4:] = M.start_values(M.GLOBAL7_TAUS)
        one('cfg3', 'global fit: 64 datasets x 1e5 pts, 4 local + 3 shared params (dim 259, block Jacobian)', trace_model(M.model_global7, 7), xs, ys,
            [1 / s for s in ss], pars, list(range(7)), [0, 0, 0, 0, 1, 1, 1], 5, 'gfh_k_sweep_gram (7 columns per dataset)', 32 + 8 * 7, 5)
    if only in (None, 4):
        from scipy.special import gammainc, gamma
        from tests.golden import goldens as G
        n = 1_000_000
        a, b = 7.5, 0.8
        xq = 0.05 + (10.0 - 0.05) * (np.arange(n) + 0.5) / n
        # data from the closed form pi/2 b^(-(a+1)/2) gamma_lower((a+1)/2, b x^2) + noise (1e6 host quadratures would take minutes)
        fq = np.pi * 0.5 * b ** (-(a + 1) / 2) * gamma((a + 1) / 2) * gammainc((a + 1) / 2, b * xq * xq)
        sq = 0.01 * (1 + np.abs(fq))
        yq = fq + sq * M.normal(n, M.SEED)
        t = trace_model(G.model_integral_single, 2); t.set_integration(rel_error=1e-10)

        def cfg4_roofline(ctx, ms, n_pts):
            """Two floors of the bisecting sweep, both FP64 VALU issue (one wave-level FP64 instruction holds its SIMD 4 cycles):
            (1) `frac`: the kernel's OWN instruction count -- SQ_ACTIVE_INST_VALU per launch from THIS round's counter pass of the
                current sources (profiles/r06_cfg4_pmc.json, tools/pmc_cfg4_r06.sh), refused when the sha1 of the generated
                translation unit differs from the one the count was taken on;
            (2) `flops_frac`: the ALGORITHM's count -- integrand evaluations of the reference's scheme for the meshes this very launch
                built (the device's own bisection records, gfh_debug_mesh_stats: an integral that ends on n intervals = 15 (2n - 1)
                evaluations on values + 15 n through AD, numerical_integration.F90:236-284, 636-664) x the necessary instructions of
                one evaluation (integrand_instr above + 3 of the rule per node), against the same issue rate = the 78.6 TFLOP/s
                peak at 2 flop per instruction and lane."""
            import hashlib
            r = {'bound': 'fp64 valu issue'}
            sha = hashlib.sha1(ctx.model_source([0, 1]).encode()).hexdigest()
            form = 'pool' if os.environ.get('GADFIT_HIP_WS_FAST') == '0' else 'scratch'
            try:
                pj = json.load(open(os.path.join(ROOT, CFG4_PMC_FILE)))
                rec = pj['forms'][form]['sweep_bisecting']
                if rec.get('source_sha1') != sha:
                    r.update(floor_ms=None, frac=None, floor_source='%s is STALE: counted on source %s, the kernel that ran is %s -- re-run tools/pmc_cfg4_r06.sh'
                                                                   % (CFG4_PMC_FILE, str(rec.get('source_sha1'))[:12], sha[:12]))
                else:
                    floor = 1e3 * rec['SQ_ACTIVE_INST_VALU'] / FP64_WAVE_INSTR_PER_S
                    r.update(floor_ms=floor, frac=floor / ms, wave_instructions_per_launch=rec['SQ_ACTIVE_INST_VALU'], source_sha1=sha,
                             floor_source='%s: %.4g wave-level FP64 VALU instructions per launch (SQ_ACTIVE_INST_VALU, rocprofv3 --pmc pass of this round on '
                                          'the kernel of this sha1, %s form) x 4 cycles / (1024 SIMDs x 2.4 GHz)' % (CFG4_PMC_FILE, rec['SQ_ACTIVE_INST_VALU'], form))
            except (OSError, ValueError, KeyError) as ex:
                r.update(floor_ms=None, frac=None, floor_source='%s not readable (%r)' % (CFG4_PMC_FILE, ex))
            try:
                ms_ = ctx.mesh_stats()
                integrals, bis = ms_['integrals'], ms_['bisections']
                iv, ig = integrand_instr(t, 1)
                rule = 15
                ev_val = rule * (integrals + 2 * bis)              # (2n - 1) panels per integral, n = bisections + 1
                ev_ad = rule * (integrals + bis)                   # n panels of the final pass
                per_val, per_ad = iv + 3, iv + ig + 3 + 2          # + node abscissa, Kronrod and Gauss sums; + the 2 weighted gradient sums
                lane_instr = ev_val * per_val + ev_ad * per_ad
                alg_ms = 1e3 * (lane_instr / 64.0) / FP64_WAVE_INSTR_PER_S
                r.update(flops_frac=alg_ms / ms, algorithmic_floor_ms=alg_ms,
                         algorithmic={'integrals': integrals, 'bisections': bis, 'mean_intervals': (integrals + bis) / max(1, integrals),
                                      'unrecorded': ms_['unrecorded'], 'evaluations_on_values': ev_val, 'evaluations_through_ad': ev_ad,
                                      'instr_per_evaluation_value': per_val, 'instr_per_evaluation_ad': per_ad,
                                      'necessary_flop_per_launch': 2.0 * lane_instr, 'fp64_peak_TFLOPs': 78.6,
                                      'achieved_TFLOPs_necessary': 2.0 * lane_instr / (ms * 1e-3) / 1e12,
                                      'source': 'mesh records of the launch before the timed ones (same parameters, same meshes), gfh_debug_mesh_stats; '
                                                'cost table: bench.py integrand_instr'})
            except Exception as ex:
                r.update(flops_frac=None, algorithmic={'error': repr(ex)[:200]})
            return r
        one('cfg4', 'pi*int_0^x t^a exp(-b t^2) dt through adaptive GK15 (rel 1e-10), N=1e6, 2 active params', t, [xq], [yq], [1.0 / sq],
            np.array([[a * 1.05, b * 0.95]]), [0, 1], [0, 0], 4, 'gfh_k_sweep (bisecting, gradient carried)', 0, 4, roofline_fn=cfg4_roofline,
            n_reps=max(3, reps // 20))
    return out


def setup_leg(_lib, M, trace_model, truth, x, y, sigma, count, active, is_global, start):
    """host clock, ms: from nothing to the end of a first fit of FIT_ITERS iterations (kernel cache warm, as after build())"""
    import subprocess
    import numpy as np
    t = [time.perf_counter()]

    def lap():
        t.append(time.perf_counter()); return 1e3 * (t[-1] - t[-2])
    # (first: what a NEW model costs before its first pass -- the headline model's kernels compiled by hiprtc into an EMPTY cache
    # directory, stored and not stored, on a compile-only context; the in-tree cache that build() fills is what every other number of
    # this leg loads from)
    import shutil, tempfile
    cold = {}
    keep_cache = os.environ.get('GADFIT_HIP_CACHE')
    tmpc = tempfile.mkdtemp(prefix='gfh_cold_cache_')
    try:
        os.environ['GADFIT_HIP_CACHE'] = tmpc
        cc = _lib.Context(-1)
        cc.set_model(trace_model(M.model_gauss8, 32))
        t0c = time.perf_counter(); cc.model_source(active); cold['generate_source_ms'] = 1e3 * (time.perf_counter() - t0c)
        t0c = time.perf_counter(); cc.model_prepare(active); cold['all_forms_ms'] = 1e3 * (time.perf_counter() - t0c)
        n_obj = len([f for f in os.listdir(tmpc) if f.endswith('.hsaco')])
        cold['code_objects'] = n_obj
        cold['per_store_form_ms'] = cold['all_forms_ms'] / 3.0
        cold['note'] = ('hiprtc compilation of the 32-parameter model into an EMPTY cache directory on a compile-only context, which compiles the three '
                        'store forms a GPU box may ask for (Jacobian stored / not stored / not stored and no residual store); a fit compiles the one '
                        'form it runs: kernels_cold_compile_ms = a third of the total')
        cc.close()
    except Exception as e:
        cold = {'error': repr(e)}
    finally:
        if keep_cache is None:
            os.environ.pop('GADFIT_HIP_CACHE', None)
        else:
            os.environ['GADFIT_HIP_CACHE'] = keep_cache
        shutil.rmtree(tmpc, ignore_errors=True)
    t[-1] = time.perf_counter()
    ctx = _lib.Context(0); ms_ctx = lap()
    tape = trace_model(M.model_gauss8, 32); ms_trace = lap()
    ctx.set_model(tape); ms_model = lap()
    ctx.set_keep_jacobian(2)                      # what the gadf_fit layers ask for: J is stored only for the fits that read it back
    ctx.model_prepare(active); ms_kernels = lap()
    ctx.set_data_begin(x, y, sigma, [0, count]); ms_begin = lap()
    ctx.init_weights(4); ms_upload = lap()        # (waits for the upload)
    _, r = ctx.fit(start, active, is_global, lambda_=1.0, max_iter=FIT_ITERS); ms_fit1 = lap()
    _, r2 = ctx.fit(start, active, is_global, lambda_=1.0, max_iter=FIT_ITERS); ms_fit2 = lap()
    # with the Jacobian kept (C-ABI default): the first fit allocates the 2.6 GB buffer; once 48 sweeps have written it the library
    # takes the job to be a long one and places the buffer (candidates timed, gfh_set_placement_tries / _after) -- in the fifth fit here
    ctx.set_keep_jacobian(1)
    _, r3 = ctx.fit(start, active, is_global, lambda_=1.0, max_iter=FIT_ITERS); ms_fit_j1 = lap()
    _, r4 = ctx.fit(start, active, is_global, lambda_=1.0, max_iter=FIT_ITERS); ms_fit_j2 = lap()
    ms_more = []
    while not ctx.placement() and len(ms_more) < 8:
        ctx.fit(start, active, is_global, lambda_=1.0, max_iter=FIT_ITERS); ms_more.append(lap())
    place = ctx.placement(); copy_rate_lib = ctx.placement_copy_GBps()
    ctx.close()
    res = {'context_ms': ms_ctx, 'trace_model_ms': ms_trace, 'set_model_ms': ms_model, 'kernels_from_cache_ms': ms_kernels,
           'kernels_cold_compile_ms': cold.get('per_store_form_ms'), 'kernels_cold_compile': cold,
           'upload_ms': ms_begin + ms_upload, 'upload_note': 'gfh_set_data_begin + wait: 240 MB from pageable host arrays, allocations, pad fill, weights',
           'first_fit_ms': ms_fit1, 'first_fit_ms_per_iteration': ms_fit1 / max(1, r.iterations),
           'second_fit_ms': ms_fit2, 'iterations_per_fit': r.iterations,
           'first_fit_keeping_the_jacobian_ms': ms_fit_j1, 'second_fit_keeping_the_jacobian_ms': ms_fit_j2,
           'later_fits_keeping_the_jacobian_ms': ms_more,
           'later_fits_note': 'the last of them contains the placement of the Jacobian buffer (after 48 sweeps on it)',
           'jacobian_placement_ms': place, 'placement_copy_GBps': copy_rate_lib,
           'to_end_of_first_fit_ms': ms_ctx + ms_trace + ms_model + ms_kernels + ms_begin + ms_upload + ms_fit1}
    exe = os.path.join(ROOT, 'tests', 'fortran', 'build', 'bench_headline')
    for key, verify in (('fortran_api', 'all'), ('fortran_api_sampled_capture', 'sample')) if os.path.exists(exe) else ():
        p = subprocess.run([exe, str(count), str(FIT_ITERS)], capture_output=True, text=True, timeout=600,
                           env=dict(os.environ, GADFIT_HIP_SETUP_TIMES='1', GADFIT_HIP_VERIFY=verify))
        f = {}
        for ln in (p.stdout + p.stderr).splitlines():
            if ln.startswith('gadf_init + add_dataset'):
                f['gadf_init_add_dataset_set_ms'] = float(ln.split(':')[1].split()[0])
            elif ln.startswith('first gadf_fit'):
                f['first_gadf_fit_ms'] = float(ln.split(':')[1].split()[0])
            elif ln.startswith('gadf_fit    '):
                f['later_gadf_fit_ms'] = float(ln.split(':')[1].split()[0])
            elif ln.startswith('gadf_fit [ms]:') and 'phases_of_first_gadf_fit' not in f:
                f['phases_of_first_gadf_fit'] = ln[len('gadf_fit [ms]:'):].strip()
        f['note'] = ('tests/fortran/bench_headline.F90: the same workload through gadf_init / gadf_add_dataset / gadf_set / gadf_fit; the first '
                     'gadf_fit records eval() ' + ('at EVERY abscissa of the data (the default: what the reference, which evaluates eval() afresh at every point, would see)'
                                                   if verify == 'all' else 'over 2^17 evenly spaced abscissas (GADFIT_HIP_VERIFY=sample: for an eval() known to treat x through AD arithmetic only)') +
                     ' while a library thread uploads the points')
        res[key] = f if p.returncode == 0 else {'error': (p.stdout + p.stderr)[-400:]}
    return res
