import sys, os, faulthandler
faulthandler.enable(); faulthandler.dump_traceback_later(60, exit=True)
R=os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, R); sys.path.insert(0, R+'/tests')
import fixtures as fx, sage2_amd as s2, numpy as np
def P(*a): print(*a, flush=True)
name, world = "g2_clean150_k40", 2
m = fx.golden(name); bases, off = fx.make_reads(m["synth"])
P("import torch"); import torch
from sage2_amd import dist as sd
P("torch", torch.__version__, torch.cuda.is_available())
dev = torch.device("cuda", 0)
ctxs = []
for r in range(world):
    c = s2.Context(m["k"], device=0, rank=r, world=world); P("ctx", r)
    c.reads_add_ascii(bases, off); c.reads_organize(); c.index_build(); P("index", r); c.overlap_probe_shard(); P("probe", r)
    ctxs.append(c)
n = ctxs[0].reads_stats().unique_reads; ms = sd.max_shard(n, world)
sends = []
for c in ctxs:
    t = torch.zeros(ms * sd.RECORD_BYTES, dtype=torch.uint8, device=dev); P("zeros")
    c.shard_export_records(t.data_ptr(), ms); sends.append(t); P("export")
P("done")
