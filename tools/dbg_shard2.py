import sys, os, faulthandler
faulthandler.enable(); faulthandler.dump_traceback_later(90, exit=True)
R=os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, R); sys.path.insert(0, R+'/tests')
import fixtures as fx, sage2_amd as s2, numpy as np
def P(*a): print(*a, flush=True)
m = fx.golden("g2_clean150_k40"); bases, off = fx.make_reads(m["synth"])
c = s2.Context(m["k"]); c.reads_add_ascii(bases, off); c.reads_organize(); c.run_steps23(); P("first ctx done", c.overlap_stats().edges); c.close()
P("import torch"); import torch
P("torch", torch.__version__); P(torch.cuda.is_available())
dev = torch.device("cuda", 0)
t = torch.zeros(10, device=dev); P("zeros ok"); torch.cuda.synchronize(); P("sync ok")
c = s2.Context(m["k"], device=0); c.reads_add_ascii(bases, off); c.reads_organize(); c.run_steps23(); P("second ctx done")
