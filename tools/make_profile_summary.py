"""Condense rocprofv3 outputs under gpurun_out/ into the tracked files under profiles/.

usage: python tools/make_profile_summary.py <round-tag> <stats_dir> <bench_json> <pmc_dir>...
  kernel stats csv -> profiles/<tag>_kernel_stats.csv, the bench line of the same profiled run -> profiles/<tag>_bench_under_rocprof.json,
  PMC passes (one counter group per directory) -> profiles/<tag>_pmc_summary.txt and profiles/probe_traffic.json
FETCH_SIZE / WRITE_SIZE are reported by rocprofv3 in KiB (guide: MI355X_MICROARCH.md, HBM section).  Every read request of the probe
kernel is a 128-byte line (TCC_EA0_RDREQ_128B == TCC_EA0_RDREQ) which FETCH_SIZE tallies at 64 bytes: reads = requests x 128 B.
"""
import csv, glob, json, os, shutil, sys, collections

tag, stats_dir, bench_json, pmc_dirs = sys.argv[1], sys.argv[2], sys.argv[3], sys.argv[4:]
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, root)
def csrc_sha1():      # the same hash bench.py checks the records against (the whole device source: kernels and launchers)
    import hashlib
    d = os.path.join(root, "sage2_amd", "csrc")
    names = sorted(glob.glob(os.path.join(d, "kernels_*.inc")) + [os.path.join(d, "sage2ov_device.hip"), os.path.join(d, "sage2ov_internal.h")])
    h = hashlib.sha1()
    for nm in names:
        h.update(os.path.basename(nm).encode()); h.update(open(nm, "rb").read())
    return h.hexdigest()
prof = os.path.join(root, "profiles")
ks = glob.glob(os.path.join(stats_dir, "**", "*kernel_stats.csv"), recursive=True)[0]
shutil.copy(ks, os.path.join(prof, f"{tag}_kernel_stats.csv"))
line = [l for l in open(bench_json) if l.startswith("{")][-1]
open(os.path.join(prof, f"{tag}_bench_under_rocprof.json"), "w").write(line)
bench = json.loads(line)
N = bench["config"]["unique_reads"]
acc = collections.defaultdict(float); cnt = collections.defaultdict(int)
for d in pmc_dirs:
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            k = (r["Kernel_Name"].split("(")[0], r["Counter_Name"]); acc[k] += float(r["Counter_Value"]); cnt[k] += 1
with open(os.path.join(prof, f"{tag}_pmc_summary.txt"), "w") as o:
    o.write(f"kernel counter value_per_launch launches per_read(N={N})\n")
    for (k, c), v in sorted(acc.items(), key=lambda kv: (kv[0][1], kv[0][0])):
        if not k.startswith(("s2::", "void s2::")):
            continue
        n = cnt[(k, c)]; o.write(f"{k} {c} {v / n:.6g} {n} {v / n / N:.3f}\n")
# the probe pass of a step is up to three launches of k_probe_fast<..., HITS = 0, TAIL> (sample, rest of the range, listed reads): the figures
# are per PASS = summed over those launches, divided by the number of steps profiled (= launches of k_ix_window, one per index build)
import re
def is_pass_kernel(k):
    m = re.search(r"k_probe_fast<([^>]*)>", k)
    return bool(m) and [x.strip() for x in m.group(1).split(",")][4] == "0"
pk = sorted({k for (k, c) in acc if is_pass_kernel(k)})
if pk:
    k = " + ".join(pk)
    def per_launch(c):
        passes = max(cnt.get(("s2::k_ix_window", c), 0), 1)
        return sum(acc.get((kk, c), 0) for kk in pk) / passes
    f = per_launch("FETCH_SIZE"); w = per_launch("WRITE_SIZE")
    # request sizes at the L2's memory side: on gfx950 every read request of this kernel is a 128-byte line (TCC_EA0_RDREQ_128B == TCC_EA0_RDREQ),
    # which FETCH_SIZE tallies at 64 bytes -- the guide's "double it" case; writes are 32-byte requests and WRITE_SIZE counts them exactly
    r128, r64, r32 = per_launch("TCC_EA0_RDREQ_128B_sum"), per_launch("TCC_EA0_RDREQ_64B_sum"), per_launch("TCC_EA0_RDREQ_32B_sum")
    rd_bytes = (128.0 * r128 + 64.0 * r64 + 32.0 * r32) if r128 else 2.0 * f * 1024.0
    if f:
        tj = os.path.join(prof, "probe_traffic.json")
        cur = json.load(open(tj)) if os.path.exists(tj) else {}
        import re
        m = re.match(r"(\d+) x (\d+) bp .*?k=(\d+).*?seed (\d+)", bench["config"]["workload"])
        key = f"{m.group(1)}x{m.group(2)}_k{m.group(3)}_seed{m.group(4)}"
        src_sha = csrc_sha1()
        cur[key] = {"kernel_source_sha1": src_sha,     # bench.py reports this traffic only while the kernel's source is the one that was profiled
                    "bytes_per_launch": rd_bytes + w * 1024.0, "read_bytes": rd_bytes, "write_bytes": w * 1024.0, "fetch_KiB_as_reported": f, "write_KiB": w,
                    "read_requests_128B": r128, "kernel": k, "source": f"profiles/{tag}_pmc_summary.txt",
                    "note": "per probe pass (sum over the pass's launches of the kernel); reads = TCC_EA0_RDREQ_128B x 128 B (all read requests are 128-byte lines; FETCH_SIZE tallies them at 64 B, cf. MI355X_MICROARCH.md HBM section)"}
        json.dump(cur, open(tj, "w"), indent=1)
# instruction counts of the probe pass per read (SQ_INSTS_* are wave-instructions): what bench.py's roofline.valu is computed from
if pk and any(c == "SQ_INSTS_VALU" for (_, c) in acc):
    import hashlib, re
    def per_read(c):
        passes = max(cnt.get(("s2::k_ix_window", c), 0), 1)
        return sum(acc.get((kk, c), 0) for kk in pk) / passes / N
    m = re.match(r"(\d+) x (\d+) bp .*?k=(\d+).*?seed (\d+)", bench["config"]["workload"])
    key = f"{m.group(1)}x{m.group(2)}_k{m.group(3)}_seed{m.group(4)}"
    ij = os.path.join(prof, "probe_insts.json")
    cur = json.load(open(ij)) if os.path.exists(ij) else {}
    src_sha = csrc_sha1()
    cur[key] = {"kernel_source_sha1": src_sha, "valu_per_read": per_read("SQ_INSTS_VALU"), "salu_per_read": per_read("SQ_INSTS_SALU"), "lds_per_read": per_read("SQ_INSTS_LDS"),
                "vmem_rd_per_read": per_read("SQ_INSTS_VMEM_RD"), "wave_quad_cycles_per_read": per_read("SQ_WAVE_CYCLES"), "kernel": " + ".join(pk), "source": f"profiles/{tag}_pmc_summary.txt",
                "note": "wave-instructions of the probe pass (all launches of k_probe_fast<..., HITS = 0, .>) per unique read, rocprofv3 --pmc SQ_INSTS_*"}
    json.dump(cur, open(ij, "w"), indent=1)
# memory-side bytes of the index build per step (the second-largest phase): every kernel of dev_build_index, reads = TCC_EA0_RDREQ x 128 B (or FETCH_SIZE x 2), writes = WRITE_SIZE.
# k_pt_hist / k_scan_* also serve convert's three passes (3 of the 9 k_pt_hist launches of a step); k_pt_scatter<4, 0> is convert's alone and is left out
if any(c == "WRITE_SIZE" for (_, c) in acc):
    import re
    steps = max(cnt.get(("s2::k_ix_window", "WRITE_SIZE"), 0), 1)
    def is_index(k):
        return any(x in k for x in ("k_minimizer", "k_pt_hist", "k_pt_scatter<3", "k_loc_index", "k_loc_scatter", "k_ix_tuples", "k_pt_bounds", "k_ix_window", "k_index_purity"))
    per = {}
    for (k, c), v in acc.items():
        if not is_index(k): continue
        share = 6.0 / 9.0 if "k_pt_hist" in k else 1.0
        e = per.setdefault(k, {"read": 0.0, "write": 0.0})
        if c == "TCC_EA0_RDREQ_128B_sum": e["read"] = v * 128.0 * share / steps
        elif c == "FETCH_SIZE" and not acc.get((k, "TCC_EA0_RDREQ_128B_sum")): e["read"] = v * 2048.0 * share / steps
        elif c == "WRITE_SIZE": e["write"] = v * 1024.0 * share / steps
    tot = sum(e["read"] + e["write"] for e in per.values())
    m = re.match(r"(\d+) x (\d+) bp .*?k=(\d+).*?seed (\d+)", bench["config"]["workload"])
    key = f"{m.group(1)}x{m.group(2)}_k{m.group(3)}_seed{m.group(4)}"
    xj = os.path.join(prof, "index_traffic.json")
    cur = json.load(open(xj)) if os.path.exists(xj) else {}
    cur[key] = {"kernel_source_sha1": csrc_sha1(), "bytes_per_step": tot, "per_kernel_GB": {k.replace("void ", "").replace("s2::", "")[:40]: round((e["read"] + e["write"]) / 1e9, 3) for k, e in sorted(per.items(), key=lambda kv: -(kv[1]["read"] + kv[1]["write"]))},
                "source": f"profiles/{tag}_pmc_summary.txt", "note": "memory-side bytes of the index build per step: reads = TCC_EA0_RDREQ_128B x 128 B, writes = WRITE_SIZE; 6 of the 9 k_pt_hist launches of a step (the other 3 are convert's)"}
    json.dump(cur, open(xj, "w"), indent=1)
print("ok")
