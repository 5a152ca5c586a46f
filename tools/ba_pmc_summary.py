"""Per-kernel table of the four PMC passes of tools/run_ba_pmc.sh (one lockstep batch of 64 C4-size LocalBA problems, per-launch means):
VALU instructions per launch, VALU busy fraction (x 4 cycles / 1024 SIMDs / kernel cycles), waves per SIMD, wait fraction, HBM bytes.
usage: python tools/ba_pmc_summary.py [out.json]"""
import sqlite3, collections, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def load(tag):
    db = os.path.join(ROOT, "gpurun_out", "bapmc_" + tag, "run_results.db")
    c = sqlite3.connect(db)
    acc = collections.defaultdict(lambda: collections.defaultdict(lambda: [0.0, set()]))
    for did, name, cn, val in c.execute("select dispatch_id,kernel_name,counter_name,value from counters_collection"):
        k = name.split("(")[0].replace("orbhip::", "").replace("void ", "")
        a = acc[k][cn]; a[0] += val; a[1].add(did)
    return {k: dict({cn: v[0] / len(v[1]) for cn, v in d.items()}, _launches=len(set().union(*[v[1] for v in d.values()]))) for k, d in acc.items()}


ins, act, fe, wr = load("insts"), load("active"), load("fetch"), load("write")
out = {"note": "rocprofv3 --pmc passes over tools/ba_batch_thr.py 64:1:1 (64 C4-size LocalBA problems in one lockstep batch, experiments build, "
               "ORBHIP_BA_GRAPH=0), per-launch means; kernel cycles = GRBM_GUI_ACTIVE / 8 XCDs; FETCH/WRITE in KB", "kernels": {}}
rows = []
for k, i in ins.items():
    if not k.startswith("k_") or k not in act: continue
    a = act[k]; cyc = a["GRBM_GUI_ACTIVE"] / 8.0
    d = {"launches": i["_launches"], "waves": i["SQ_WAVES"], "valu_insts": i["SQ_INSTS_VALU"], "valu_busy_frac": i["SQ_INSTS_VALU"] * 4.0 / 1024.0 / cyc,
         "vmem_rd_insts": i["SQ_INSTS_VMEM_RD"], "vmem_wr_insts": i["SQ_INSTS_VMEM_WR"], "lds_insts": i["SQ_INSTS_LDS"],
         "kernel_us": cyc / 2400.0, "avg_waves_per_simd": a["SQ_WAVE_CYCLES"] * 4.0 / 1024.0 / cyc, "wait_any_frac": a["SQ_WAIT_ANY"] / max(a["SQ_WAVE_CYCLES"], 1.0),
         "busy_cu_frac": a.get("SQ_BUSY_CU_CYCLES", 0.0) * 4.0 / 256.0 / cyc / 4.0,
         "lds_bank_conflict_cycles": a.get("SQ_LDS_BANK_CONFLICT", 0.0),
         "fetch_KB": fe.get(k, {}).get("FETCH_SIZE"), "write_KB": wr.get(k, {}).get("WRITE_SIZE")}
    out["kernels"][k] = d
    rows.append((d["kernel_us"] * d["launches"], k, d))
rows.sort(reverse=True)
print("%-24s %5s %9s %8s %7s %7s %7s %9s %9s %8s" % ("kernel", "calls", "us/launch", "waves", "valu%", "wv/simd", "wait%", "fetch MB", "write MB", "GB/s"))
for _, k, d in rows[:16]:
    mb_f = (d["fetch_KB"] or 0) / 1024.0; mb_w = (d["write_KB"] or 0) / 1024.0
    print("%-24s %5d %9.1f %8d %7.1f %7.2f %7.1f %9.1f %9.1f %8.0f" % (k[:24], d["launches"], d["kernel_us"], d["waves"], 100 * d["valu_busy_frac"], d["avg_waves_per_simd"],
                                                                       100 * d["wait_any_frac"], mb_f, mb_w, (mb_f + mb_w) / 1024.0 / (d["kernel_us"] * 1e-6)))
if len(sys.argv) > 1:
    json.dump(out, open(sys.argv[1], "w"), indent=1)
