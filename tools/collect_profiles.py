"""Copy the evidence tools/run_profiles.sh left under gpurun_out/ (scratch) into profiles/ (tracked), named per round.
usage: python tools/collect_profiles.py r02"""
import csv, json, os, shutil, sqlite3, subprocess, sys, collections
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
RND = sys.argv[1] if len(sys.argv) > 1 else "r02"
G = os.path.join(ROOT, "gpurun_out"); O = os.path.join(G, RND); P = os.path.join(ROOT, "profiles")

def first_json_line(path):
    for line in open(path):
        line = line.strip()
        if line.startswith("{"):
            return json.loads(line)
    raise SystemExit("no JSON line in " + path)

json.dump(first_json_line(os.path.join(O, "bench.json")), open(os.path.join(P, RND + "_bench.json"), "w"), indent=1)
json.dump(first_json_line(os.path.join(O, "bench_2rank_shared.json")), open(os.path.join(P, RND + "_bench_2rank_shared_gpu.json"), "w"), indent=1)
shutil.copy(os.path.join(O, "bench_kernel_stats.csv"), os.path.join(P, RND + "_bench_kernel_stats.csv"))
shutil.copy(os.path.join(O, "localba_batch16_kernel_stats.csv"), os.path.join(P, RND + "_localba_batch16_kernel_stats.csv"))
old = json.load(open(os.path.join(P, RND + "_api_latency.json"))) if os.path.exists(os.path.join(P, RND + "_api_latency.json")) else {}
api = {"note": old.get("note", "wall latency of the per-frame host-pointer entry points on one MI355X (C++ through the C ABI, tools/cpp/api_latency.cpp; Python ctypes mirror, tools/api_latency.py)"),
       "cpp": first_json_line(os.path.join(O, "api_latency_cpp.json")), "python": first_json_line(os.path.join(O, "api_latency_py.json"))}
if "round1_python" in old: api["round1_python"] = old["round1_python"]
json.dump(api, open(os.path.join(P, RND + "_api_latency.json"), "w"), indent=1)
txt = open(os.path.join(O, "fast_phase_prof.json")).read().strip()
open(os.path.join(P, RND + "_fast_phase_prof.json"), "w").write(txt + "\n")
# raw per-dispatch PMC tables (rows of one dispatch summed over its shader engines) + the two summaries
for tag in ("insts", "active", "fetch", "write"):
    db = os.path.join(G, "pmc_x_" + tag, "run_results.db")
    c = sqlite3.connect(db)
    acc = collections.OrderedDict(); names = set()
    for did, name, cn, val in c.execute("select dispatch_id,kernel_name,counter_name,value from counters_collection order by dispatch_id"):
        k = name.split("(")[0].replace("orbhip::", "").replace("void ", "")
        acc.setdefault(did, [k, collections.defaultdict(float)])[1][cn] += val; names.add(cn)
    names = sorted(names)
    with open(os.path.join(P, "%s_pmc_%s_counter_collection.csv" % (RND, tag)), "w", newline="") as f:
        w = csv.writer(f); w.writerow(["dispatch_id", "kernel"] + names)
        for did, (k, d) in acc.items():
            w.writerow([did, k] + [d.get(n, 0.0) for n in names])
subprocess.check_call([sys.executable, os.path.join(ROOT, "tools", "pmc_extract.py"), RND])
print("profiles/%s_* refreshed" % RND)
