"""diagnostic (build container, CPU): the step-4 restatement against the reference's own classes (oracle/_ref) on many random graphs -- the
same generator settings as step4_stress.py uses on the GPU"""
import ctypes, os, sys, tempfile
R=os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0,R); sys.path.insert(0,R+'/tests')
import numpy as np, graphgen as gg
lib = ctypes.CDLL(R+"/oracle/liboracle_step4.so"); lib.orc4_run_files.argtypes=[ctypes.c_char_p, ctypes.c_ulonglong, ctypes.c_char_p, ctypes.POINTER(ctypes.c_ulonglong)]
drv = ctypes.CDLL(R+"/oracle/_ref/libsage2ref_driver.so"); drv.sage2ref_run_step4.argtypes=[ctypes.c_char_p, ctypes.c_int, ctypes.c_int, ctypes.c_char_p, ctypes.POINTER(ctypes.c_double), ctypes.POINTER(ctypes.c_ulonglong)]
tmp=tempfile.mkdtemp(); bad=0; n=0
NMAX=6000; gg.write_reads(tmp+"/t.reads", NMAX, read_len=60, seed=1)
for seed in range(int(sys.argv[1]), int(sys.argv[2])):
    r=np.random.default_rng(seed)
    N,e=gg.random_graph(seed, n_anchor=int(r.integers(1,60)), n_paths=int(r.integers(1,150)), max_len=int(r.integers(0,25)), n_cycles=int(r.integers(0,8)), p_bad=float(r.choice([0,0.02,0.1])), len_hi=int(r.choice([3,12,25,80])))
    if N>NMAX: continue
    gg.write_graph3(tmp+"/t.graph3", NMAX, e, read_len=60)
    t=(ctypes.c_double*2)(); c=(ctypes.c_ulonglong*4)(); c2=(ctypes.c_ulonglong*5)()
    assert drv.sage2ref_run_step4((tmp+"/t").encode(), 40, 1, (tmp+"/ref4").encode(), t, c)==0
    assert lib.orc4_run_files((tmp+"/t.graph3").encode(), NMAX, (tmp+"/orc4").encode(), c2)==0
    ok = open(tmp+"/ref4","rb").read()==open(tmp+"/orc4","rb").read() and (c[1],c[2],c[3])==(c2[1],c2[2],c2[3])
    n+=1
    if not ok: bad+=1; print("MISMATCH", seed, N, len(e), flush=True)
print("graphs", n, "mismatches", bad)
