"""Rebuild profiles/<round>_pmc_valu.json and <round>_pmc_traffic.json (round = argv[1]) from the four rocprofv3 PMC passes
of tools/run_pmc.sh over tools/frontend_only.py (rocpd databases gpurun_out/pmc_x_{insts,active,fetch,write}, all at the
batch size written to gpurun_out/pmc_x_batch.txt: the bench's 256 frames).  Nothing under profiles/ is touched unless all
four databases hold every kernel of the front-end."""
import sqlite3, collections, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
RND = sys.argv[1] if len(sys.argv) > 1 else "r05"
MATCH = os.environ.get("PMC_MATCH_KERNEL", "k_match_pairs_mfma")      # (k_match_pairs when the passes ran with ORBHIP_MATCH_MFMA=0)
BLUR = os.environ.get("PMC_BLUR_KERNEL", "k_blur7_mfma")      # (k_blur7 when the passes ran with ORBHIP_BLUR_MFMA=0)
RESIZE = os.environ.get("PMC_RESIZE_KERNEL", "k_resize_mfma")  # (k_resize when the passes ran with ORBHIP_RESIZE_MFMA=0)
NAMES = [RESIZE, "k_fast_cells", "k_octree", BLUR, "k_describe", MATCH]


def load(tag):
    db = os.path.join(ROOT, "gpurun_out", "pmc_x_" + tag, "run_results.db")
    if not os.path.exists(db) or os.path.getsize(db) == 0:
        raise SystemExit("pmc_extract: %s missing or empty" % db)
    c = sqlite3.connect(db)
    acc = collections.defaultdict(lambda: collections.defaultdict(lambda: [0.0, set()]))
    try:
        rows = list(c.execute("select dispatch_id,kernel_name,counter_name,value from counters_collection"))
    except sqlite3.OperationalError as e:
        raise SystemExit("pmc_extract: %s holds no counters (%s)" % (db, e))
    for did, name, cn, val in rows:
        k = name.split("(")[0].replace("orbhip::", "").replace("void ", "").split("<")[0]
        a = acc[k][cn]; a[0] += val; a[1].add(did)
    out = {k: {cn: v[0] / len(v[1]) for cn, v in d.items()} for k, d in acc.items()}      # mean per launch (rows of one dispatch summed)
    for k in d_launches(acc):
        out[k]["_launches"] = d_launches(acc)[k]
    missing = [k for k in NAMES if k not in out]
    if missing:
        raise SystemExit("pmc_extract: %s lacks kernels %s" % (db, missing))
    return out


def d_launches(acc):
    return {k: len(set().union(*[v[1] for v in d.values()])) for k, d in acc.items()}


ins, act, fe, wr = load("insts"), load("active"), load("fetch"), load("write")
try:
    B = int(open(os.path.join(ROOT, "gpurun_out", "pmc_x_batch.txt")).read().strip())
except Exception:
    raise SystemExit("pmc_extract: gpurun_out/pmc_x_batch.txt missing (written by tools/run_pmc.sh)")
valu = {"note": "rocprofv3 --pmc passes over tools/frontend_only.py %d 3 (extract + match of ONE %d-frame batch of the bench's 1241x376 frames, "
                "2000 features, this round's kernels), --pmc only, one pass per counter group. valu_busy_frac = SQ_INSTS_VALU x 4 cycles / "
                "(1024 SIMDs x kernel cycles), kernel cycles = GRBM_GUI_ACTIVE / 8 XCDs; per-launch means; k_resize = sum of its 7 launches "
                "per batch." % (B, B),
        "frames_per_launch": B, "kernels": {}}
for k in NAMES:
    i, a = ins[k], act[k]
    launches = 7 if k in ("k_resize", "k_resize_mfma") else 1
    cyc = a["GRBM_GUI_ACTIVE"] / 8.0 * launches
    valu["kernels"][k] = {
        "waves": i["SQ_WAVES"] * launches, "valu_insts": i["SQ_INSTS_VALU"] * launches, "salu_insts": i["SQ_INSTS_SALU"] * launches,
        "lds_insts": i["SQ_INSTS_LDS"] * launches, "vmem_rd_insts": i["SQ_INSTS_VMEM_RD"] * launches, "vmem_wr_insts": i["SQ_INSTS_VMEM_WR"] * launches,
        "kernel_cycles": cyc,
        "valu_insts_per_wave": i["SQ_INSTS_VALU"] / i["SQ_WAVES"],
        "valu_busy_frac": i["SQ_INSTS_VALU"] * launches * 4.0 / 1024.0 / cyc,
        "avg_waves_per_simd": a["SQ_WAVE_CYCLES"] * launches * 4.0 / 1024.0 / cyc if "SQ_WAVE_CYCLES" in a else None,
        "wait_any_frac_of_wave_cycles": a["SQ_WAIT_ANY"] / a["SQ_WAVE_CYCLES"] if "SQ_WAVE_CYCLES" in a else None,
        "lds_bank_conflict_cycles": a.get("SQ_LDS_BANK_CONFLICT", 0.0) * launches,
    }
    if "SQ_INSTS_MFMA" in i:
        valu["kernels"][k]["mfma_insts"] = i["SQ_INSTS_MFMA"] * launches
    if "SQ_VALU_MFMA_BUSY_CYCLES" in a:            # (summed over the SIMDs: / 1024 SIMDs / kernel cycles = fraction of the matrix pipes' time)
        valu["kernels"][k]["mfma_busy_frac"] = a["SQ_VALU_MFMA_BUSY_CYCLES"] * launches / 1024.0 / cyc
    if k == MATCH:
        valu["kernels"][k]["is_match"] = True
tr = {"note": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes (tools/frontend_only.py %d 3: 3 x extract + match of one %d-frame batch, "
              "1241x376, 2000 features, this round's kernels); units KB (TCC_EA0 requests x 64 B / 1024). Calibration on gfx950 (round 1): "
              "k_blur7's 4 B/lane dword reads report ~1.2x its algorithmic read bytes (its tile halo alone predicts 1.19x), so no x2 correction "
              "is applied to these 1-4 B/lane kernels (the guide's x2 applies to 16 B/lane streams)." % (B, B),
      "frames_per_launch": B, "kernels": {}}
for k in NAMES:
    mult = 7 if k in ("k_resize", "k_resize_mfma") else 1                      # 7 pyramid launches per batch
    f, w = fe[k]["FETCH_SIZE"] * mult, wr[k]["WRITE_SIZE"] * mult
    tr["kernels"][k] = {"FETCH_SIZE_KB_per_batch": f, "WRITE_SIZE_KB_per_batch": w, "hbm_bytes_per_frame": (f + w) * 1024.0 / B}
P = os.environ.get("PROFILES_OUT") or os.path.join(ROOT, "profiles")
for name, obj in ((RND + "_pmc_valu.json", valu), (RND + "_pmc_traffic.json", tr)):
    tmp = os.path.join(P, name + ".tmp")
    json.dump(obj, open(tmp, "w"), indent=1)
    os.replace(tmp, os.path.join(P, name))
for k in NAMES:
    v = valu["kernels"][k]
    print("%-14s VALU/wave %.0f  busy %.3f  waves/SIMD %.2f  wait %.2f  HBM B/frame %.0f" % (k, v["valu_insts_per_wave"], v["valu_busy_frac"], v["avg_waves_per_simd"] or 0,
                                                                                 v["wait_any_frac_of_wave_cycles"] or 0, tr["kernels"][k]["hbm_bytes_per_frame"]))
