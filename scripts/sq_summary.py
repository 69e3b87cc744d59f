"""Derived figures from the SQ counter passes (profiles/r03_pmc_sq_group{1,2,3}.csv, one rocprofv3 --pmc pass each over
`bench.py --pmc-child`): per kernel and launch, how busy the matrix pipe, the vector ALU and the LDS were and what the
resident waves were doing.  Units (MI355X_MICROARCH.md): SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_* count quad-cycles
summed over waves; SQ_VALU_MFMA_BUSY_CYCLES counts cycles summed over SIMDs; GRBM_GUI_ACTIVE counts cycles summed over the
8 XCDs; SQ_LDS_IDX_ACTIVE / SQ_LDS_BANK_CONFLICT count cycles summed over CUs.
    python scripts/sq_summary.py profiles/r03_pmc_sq_group > profiles/r03_sq_summary.txt"""
import csv, collections, sys
prefix = sys.argv[1]
rows = collections.defaultdict(dict)
for g in (1, 2, 3):
    for r in csv.DictReader(open(f"{prefix}{g}.csv")):
        rows[r["kernel"]][r["counter"]] = float(r["mean_per_dispatch"])
CUS, SIMDS, XCDS = 256, 1024, 8
print(f"{'kernel':44s} {'ms*':>6s} {'MFMA%':>6s} {'VALU%':>6s} {'LDS%':>6s} {'conf%':>6s} | wave time: {'issue%':>7s} {'stall%':>7s} {'wait%':>6s} (of which LDS-issue {'%':>3s}) | per wave-instr: VALU SALU LDS MFMA VMEM")
for k, v in rows.items():
    if "GRBM_GUI_ACTIVE" not in v:
        continue
    cyc = v["GRBM_GUI_ACTIVE"] / XCDS                     # kernel duration in shader cycles (under the profiler)
    name = k.replace("void ", "").split("(")[0][:44]
    mfma = v.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) / (cyc * SIMDS)
    valu = 4 * v.get("SQ_ACTIVE_INST_VALU", 0) / (cyc * SIMDS)
    lds = v.get("SQ_LDS_IDX_ACTIVE", 0) / (cyc * CUS)
    conf = v.get("SQ_LDS_BANK_CONFLICT", 0) / max(v.get("SQ_LDS_IDX_ACTIVE", 1), 1)
    wc = max(v.get("SQ_WAVE_CYCLES", 1), 1)
    waves = max(v.get("SQ_WAVES", 1), 1)
    print(f"{name:44s} {cyc / 2.4e6:6.3f} {100 * mfma:6.1f} {100 * valu:6.1f} {100 * lds:6.1f} {100 * conf:6.1f} | "
          f"{100 * v.get('SQ_ACTIVE_INST_ANY', 0) / wc:18.1f} {100 * v.get('SQ_WAIT_INST_ANY', 0) / wc:7.1f} {100 * v.get('SQ_WAIT_ANY', 0) / wc:6.1f} "
          f"{100 * v.get('SQ_WAIT_INST_LDS', 0) / wc:22.1f} | "
          f"{v.get('SQ_INSTS_VALU', 0) / waves:9.0f} {v.get('SQ_INSTS_SALU', 0) / waves:5.0f} {v.get('SQ_INSTS_LDS', 0) / waves:5.0f} "
          f"{v.get('SQ_INSTS_MFMA', 0) / waves:5.0f} {(v.get('SQ_INSTS_VMEM_RD', 0) + v.get('SQ_INSTS_VMEM_WR', 0)) / waves:5.0f}")
print("\n* kernel duration under the profiler at a nominal 2.4 GHz; MFMA% = matrix-pipe busy cycles / (duration x 1024 SIMDs); VALU% = 4 x "
      "ACTIVE_INST_VALU quad-cycles / the same; LDS% = LDS-array active cycles / (duration x 256 CUs); conf% = share of LDS cycles lost to bank "
      "conflicts; issue / stall / wait = ACTIVE_INST_ANY / WAIT_INST_ANY / WAIT_ANY as shares of SQ_WAVE_CYCLES (waves issuing, stalled at issue, "
      "parked on s_waitcnt or a barrier).")
