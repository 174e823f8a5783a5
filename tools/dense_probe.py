import os, sys, ctypes as C
os.environ["MVS_USE_DEBUG_LIB"]="1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mvslam_amd import capi, synth
P=256
data = synth.make_batch(0, P, n_kp=2000)
ctx = capi.Context(0); b = capi.Batch(ctx, P, 2000, 32)
b.upload(0, data["desc1"], data["kp1"], data["n1"], data["desc2"], data["kp2"], data["n2"], data["K"], data["global_index"])
prm = capi.default_params(num_hypotheses=50000, sampler=capi.SAMPLER_PHILOX, seed=synth.SEED_BASE, max_error_sq=1e-2)
lib = capi.lib()
for dense in ((1, 1, 1) if os.environ.get('DENSE_ONLY') else (0, 1, 0, 1)):
    lib.mvs_debug_set_count_dense(C.c_int(dense))
    b.run(prm); b.sync()
    t = {}
    for n, ms in b.time_kernels(prm, steps=3):
        t[n] = t.get(n, 0) + ms
    print(dense, {k: round(v, 3) for k, v in t.items() if 'count' in k}, flush=True)
for dense in (0, 1):
    lib.mvs_debug_set_count_dense(C.c_int(dense))
    ws = b.stats(prm)
    print(dense, {k: ws[k] for k in ("score_evals", "score_evals_executed", "score_evals_executed_f32", "exact_solves", "hypotheses")}, flush=True)
for k, v in ctx.kernel_info(2000, 32).items():
    if "mfma" in k or "count32" in k:
        print(k, v, flush=True)
