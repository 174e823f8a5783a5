import os, sys, ctypes as C
os.environ["MVS_USE_DEBUG_LIB"]="1"
sys.path.insert(0,'/root/repo')
from mvslam_amd import capi, synth
import numpy as np
F=int(os.environ.get("SEQ_F","33"))
seq = synth.make_sequence(F, n_kp=2000)
P=F-1
ctx = capi.Context(0); b = capi.Batch(ctx, P, 2000, 32)
K = np.tile(seq["K"].reshape(1,9),(P,1))
b.upload(0, seq["desc"][:-1], seq["kp"][:-1], seq["n_kp"][:-1], seq["desc"][1:], seq["kp"][1:], seq["n_kp"][1:], K, np.arange(P))
lib = capi.lib()
for thr in (1e-2,):
    prm = capi.default_params(num_hypotheses=50000, sampler=capi.SAMPLER_PHILOX, seed=synth.SEED_BASE, max_error_sq=thr)
    b.run(prm); b.sync()
    res = b.download(matches=False, mask=False, points=False)["results"]
    st = b.stats(prm)
    print(thr, st)
    for p in (0,5,20):
        info=(C.c_int32*4)()
        lib.mvs_debug_read_hyp_rec(b._h, C.c_int(p), C.c_int(1), None, None, None, info)
        print(p, list(info), res[p]["n_matches"], res[p]["best_count"])
    # band statistics in forced mode 2
    lib.mvs_debug_prescreen_only(b._h, C.byref(prm), C.c_int(P), C.c_int(2))
    H=2048
    rec=np.zeros((H,10)); state=np.zeros(H,np.uint8)
    lib.mvs_debug_read_hyp_rec(b._h, C.c_int(5), C.c_int(H), rec.ctypes.data_as(C.POINTER(C.c_double)), state.ctypes.data_as(C.POINTER(C.c_ubyte)), None, None)
    print('states', np.bincount(state, minlength=4), 'band pct (state1)', np.percentile(rec[state==1,9]-thr,[10,50,90]) if (state==1).any() else None)
    for n,ms in b.time_kernels(prm, steps=2): print(n, round(ms,3))
