"""FP64-MFMA utilisation of the reduced solve from a rocprofv3 run with --kernel-trace --pmc SQ_INSTS_VALU_MFMA_MOPS_F64
(rocpd database; ORBHIP_BA_GRAPH=0 so that counters are attributed to kernels): per kernel launches, time, MFMA flops
(MOPS x 512), TFLOP/s and the fraction of the 78.6 TFLOP/s FP64 matrix peak; plus the whole factorisation.
usage: python tools/mfma_c5.py <run_results.db> [out.json]"""
import sqlite3, sys, json, collections
c = sqlite3.connect(sys.argv[1])
dur = {}
cols = [r[1] for r in c.execute("pragma table_info(kernels)")]
idc = "dispatch_id" if "dispatch_id" in cols else "id"
for did, name, s, e in c.execute("select %s,name,start,end from kernels" % idc):
    dur[did] = (name, e - s)
mops = collections.defaultdict(float); busy = collections.defaultdict(float); gui = collections.defaultdict(float)
for did, cn, val in c.execute("select dispatch_id,counter_name,value from counters_collection"):
    if cn == "SQ_INSTS_VALU_MFMA_MOPS_F64":
        mops[did] += val
    elif cn == "SQ_VALU_MFMA_BUSY_CYCLES":
        busy[did] += val
    elif cn == "GRBM_GUI_ACTIVE":
        gui[did] = max(gui[did], val)
acc = collections.defaultdict(lambda: [0, 0.0, 0.0, 0.0, 0.0])
for did, (name, d) in dur.items():
    k = name.split("(")[0].replace("orbhip::", "").replace("void ", "")
    a = acc[k]; a[0] += 1; a[1] += d; a[2] += mops.get(did, 0.0) * 512.0; a[3] += busy.get(did, 0.0); a[4] += gui.get(did, 0.0)
out = {"peak_TFLOPs": 78.6, "measured_issue_peak_TFLOPs": 72.0, "kernels": {}}     # (72: tools/ubench/mfma_f64.hip, profiles/r04_mfma_f64_ubench.txt)
tot_t = tot_f = 0.0
for k, (n, t, f, bz, ga) in sorted(acc.items(), key=lambda kv: -kv[1][1]):
    if not k.startswith("k_chol"): continue
    out["kernels"][k] = {"launches": n, "total_ms": round(t / 1e6, 3), "mfma_GFLOP": round(f / 1e9, 2), "achieved_TFLOPs": round(f / t / 1e3, 3) if t else 0.0,
                         "frac_of_peak": round(f / t / 1e3 / 78.6, 4) if t else 0.0, "frac_of_measured_issue_peak": round(f / t / 1e3 / 72.0, 4) if t else 0.0}
    if bz and ga: out["kernels"][k]["mfma_busy_cycles_per_gui_active_cycle"] = round(bz / ga, 3)      # summed over the SIMDs that ran the kernel / the launch's active cycles
    tot_t += t; tot_f += f
out["factorisation_and_substitution"] = {"total_ms": round(tot_t / 1e6, 3), "mfma_GFLOP": round(tot_f / 1e9, 2), "achieved_TFLOPs": round(tot_f / tot_t / 1e3, 3), "frac_of_peak": round(tot_f / tot_t / 1e3 / 78.6, 4)}
txt = json.dumps(out, indent=1)
print(txt)
if len(sys.argv) > 2: open(sys.argv[2], "w").write(txt + "\n")
