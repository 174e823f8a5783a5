import os, sys, ctypes as C
os.environ["MVS_USE_DEBUG_LIB"]="1"
sys.path.insert(0,'/root/repo')
from mvslam_amd import capi, synth
data = synth.make_batch(0, 1, n_kp=2000)
ctx = capi.Context(0); b = capi.Batch(ctx, 1, 2000, 32)
b.upload(0, data["desc1"], data["kp1"], data["n1"], data["desc2"], data["kp2"], data["n2"], data["K"], data["global_index"])
prm = capi.default_params(num_hypotheses=50000, sampler=capi.SAMPLER_PHILOX, seed=synth.SEED_BASE, max_error_sq=1e-2)
lib = capi.lib()
for mf in (1, 0, 1, 0):
    lib.mvs_debug_set_match_mfma(C.c_int(mf))
    tot,_ = b.time(prm, steps=20, warmup=3, per_kernel=False)
    print("match_mfma", mf, "ms", round(tot/20,4), [(n, round(ms,4)) for n,ms in b.time_kernels(prm, steps=5)], flush=True)
