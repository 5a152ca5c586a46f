"""Copy the evidence tools/run_profiles.sh left under gpurun_out/<round>/ (scratch) into profiles/ (tracked), named per round.
usage: python tools/collect_profiles.py r03

Every input is VALIDATED before anything under profiles/ changes: it must exist, be non-empty and parse (a JSON line, a
kernel-stats CSV with at least the front-end's kernels, a PMC database with counters).  An input that fails is reported and
SKIPPED - the tracked file of an earlier, good run is never overwritten by a missing, empty or broken one (round 2 lost its
headline kernel stats exactly that way).  Files are written to a temporary name and renamed.  Exit code 1 if anything was skipped."""
import csv, json, os, sqlite3, subprocess, sys, collections
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
RND = sys.argv[1] if len(sys.argv) > 1 else "r06"
G = os.path.join(ROOT, "gpurun_out"); O = os.path.join(G, RND); P = os.environ.get("PROFILES_OUT") or os.path.join(ROOT, "profiles")      # (PROFILES_OUT: collect on the GPU box into gpurun_out/, the raw databases exceed what travels back)
os.makedirs(P, exist_ok=True)
skipped = []


def put(name, text):
    tmp = os.path.join(P, name + ".tmp")
    with open(tmp, "w") as f:
        f.write(text)
    os.replace(tmp, os.path.join(P, name))
    print("  profiles/%s (%d bytes)" % (name, len(text)))


def first_json_line(path):
    if not os.path.exists(path) or os.path.getsize(path) == 0:
        raise ValueError("missing or empty")
    for line in open(path):
        line = line.strip()
        if line.startswith("{"):
            return json.loads(line)
    raise ValueError("no JSON line")


def take_json(src, dst, must_have=()):
    try:
        j = first_json_line(os.path.join(O, src))
        for k in must_have:
            if k not in j:
                raise ValueError("no key %r" % k)
        put(dst, json.dumps(j, indent=1) + "\n")
        return j
    except Exception as e:
        skipped.append("%s: %s" % (src, e)); return None


def take_kernel_csv(src, dst, must_name=()):
    path = os.path.join(O, src)
    try:
        if not os.path.exists(path) or os.path.getsize(path) == 0:
            raise ValueError("missing or empty")
        rows = list(csv.reader(open(path)))
        if len(rows) < 2 or rows[0][:2] != ["Name", "Calls"]:
            raise ValueError("not a kernel-stats table")
        names = " ".join(r[0] for r in rows[1:])
        for k in must_name:
            if k not in names:
                raise ValueError("kernel %s absent" % k)
        put(dst, open(path).read())
    except Exception as e:
        skipped.append("%s: %s" % (src, e))


print("collecting %s -> profiles/" % O)
take_json("bench.json", RND + "_bench.json", ("value", "roofline"))
take_kernel_csv("bench_kernel_stats.csv", RND + "_bench_kernel_stats.csv", ("k_fast_cells", "k_match_pairs", "k_blur7", "k_resize", "k_describe", "k_octree"))      # (substring match: k_match_pairs_mfma too)
take_json("bench_traced.json", RND + "_bench_traced.json", ("value", "kernels"))
take_kernel_csv("localba_batch64_kernel_stats.csv", RND + "_localba_batch64_kernel_stats.csv", ("k_ba_schur", "k_chol_wg"))
take_kernel_csv("gba_c5_kernel_stats.csv", RND + "_gba_c5_kernel_stats.csv", ("k_chol",))
for st in ("covis", "dense"):
    take_kernel_csv("localba_batch64_%s_kernel_stats.csv" % st, RND + "_localba_batch64_%s_kernel_stats.csv" % st, ("k_ba_schur", "k_chol_wg"))
take_json("bench_2rank_shared.json", RND + "_bench_2rank_shared_gpu.json", ("value", "collective"))
take_json("bench_8rank_shared.json", RND + "_bench_8rank_shared_gpu.json", ("value", "collective"))
take_json("bench_rccl_ws1.json", RND + "_bench_rccl_ws1.json", ("collective",))
try:
    api = {"note": "wall latency of the per-frame host-pointer entry points on one MI355X (C++ through the C ABI, tools/cpp/api_latency.cpp; "
                   "Python ctypes mirror, tools/api_latency.py)",
           "cpp": first_json_line(os.path.join(O, "api_latency_cpp.json")), "python": first_json_line(os.path.join(O, "api_latency_py.json"))}
    put(RND + "_api_latency.json", json.dumps(api, indent=1) + "\n")
except Exception as e:
    skipped.append("api_latency: %s" % e)
for src, dst, why in (("mfma_batch.json", "_mfma_localba_batch64.json", "FP64-MFMA counters (SQ_INSTS_VALU_MFMA_MOPS_F64 x 512 flops, SQ_VALU_MFMA_BUSY_CYCLES, GRBM_GUI_ACTIVE) of 64 C4-size LocalBA problems per lockstep batch: k_chol_wg, one workgroup per problem, walking the skyline - the EXECUTED matrix flops (tools/run_mfma_pmc.sh; experiments build, ORBHIP_BA_GRAPH=0)"),
                      ("mfma_single.json", "_mfma_localba_single.json", "the same counters over single C4-size LocalBA solves: the persistent, flag-linked k_chol_persist (skyline walk)"),
                      ("mfma_batch_covis.json", "_mfma_localba_batch64_covis.json", "the same counters, 64 C4-size local maps with covisibility-window tracks + a current keyframe sharing landmarks with every keyframe (synth.make_ba_graph_covis 'covis': a band of ~6 tiles under a dense last block row)"),
                      ("mfma_batch_dense.json", "_mfma_localba_batch64_dense.json", "the same counters, 64 C4-size local maps whose every keyframe pair shares landmarks: a FULL reduced system, every tile update of the factorisation runs"),
                      ("mfma_c5.json", "_mfma_gba_c5.json", "the same counters over one GlobalBA at C5 size (500 keyframes, 2994-unknown reduced system, band 3), 10 iterations: k_chol_persist walking the skyline (round 5; k_chol_persist_blk before)")):
    path = os.path.join(O, src)
    try:
        j = json.load(open(path)); j = dict(workload=why, **j)
        put(RND + dst, json.dumps(j, indent=1) + "\n")
    except Exception as e:
        skipped.append("%s: %s" % (src, e))
for txt in ("localba_throughput.txt", "track_latency.txt", "concurrency.txt", "mfma_f64_ubench.txt", "schur_phase_prof.txt", "chol_wg_phase_prof.txt",
            "localba_trace_overlap_12_callers.txt", "pt_fuse_ab.txt", "factor_ab.txt", "f64_latency.txt"):
    path = os.path.join(O, txt)
    if os.path.exists(path) and os.path.getsize(path) > 0:
        put(RND + "_" + txt, open(path).read())
    else:
        skipped.append(txt + ": missing or empty")
for extra in ("ba_batch64_pmc.json", "fast_phase_prof.json", "chol_persist_chain_c4.json", "chol_persist_chain_c5.json", "chol_phase_prof.json", "octree_phase_prof.json", "gba_c5_mfma.json", "pcie_pipeline.json", "tracking_step.json"):
    path = os.path.join(O, extra)
    if os.path.exists(path):
        try:
            txt = open(path).read().strip()
            json.loads(txt if txt.startswith("{") and "\n{" not in txt else txt.splitlines()[-1])
            put(RND + "_" + extra, txt + "\n")
        except Exception as e:
            skipped.append("%s: %s" % (extra, e))
# raw per-dispatch PMC tables (rows of one dispatch summed over its shader engines) + the two summaries
pmc_ok = True
tables = {}
for tag in ("insts", "active", "fetch", "write"):
    db = os.path.join(G, "pmc_x_" + tag, "run_results.db")
    try:
        if not os.path.exists(db) or os.path.getsize(db) == 0:
            raise ValueError("missing or empty")
        c = sqlite3.connect(db)
        acc = collections.OrderedDict(); names = set()
        for did, name, cn, val in c.execute("select dispatch_id,kernel_name,counter_name,value from counters_collection order by dispatch_id"):
            k = name.split("(")[0].replace("orbhip::", "").replace("void ", "")
            acc.setdefault(did, [k, collections.defaultdict(float)])[1][cn] += val; names.add(cn)
        if not acc:
            raise ValueError("no counter rows")
        tables[tag] = (acc, sorted(names))
    except Exception as e:
        skipped.append("pmc_x_%s: %s" % (tag, e)); pmc_ok = False
if pmc_ok:
    import io
    for tag, (acc, names) in tables.items():
        buf = io.StringIO()
        w = csv.writer(buf); w.writerow(["dispatch_id", "kernel"] + names)
        for did, (k, d) in acc.items():
            w.writerow([did, k] + [d.get(n, 0.0) for n in names])
        put("%s_pmc_%s_counter_collection.csv" % (RND, tag), buf.getvalue())
    if subprocess.call([sys.executable, os.path.join(ROOT, "tools", "pmc_extract.py"), RND]) != 0:
        skipped.append("pmc_extract failed")
if skipped:
    print("SKIPPED (tracked files of earlier runs left untouched):")
    for s in skipped:
        print("  " + s)
    sys.exit(1)
print("profiles/%s_* refreshed" % RND)
