"""Clock-under-load record of the dominant kernel from the committed counter files (run after tools/condense_r06.py):
python tools/clock_record.py > profiles/r06_clock_under_load.txt"""
import csv
import os

P = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "profiles")
pmc = {r["counter"]: float(r["avg_per_launch"]) for r in csv.DictReader(open(os.path.join(P, "r06_meanshift_x3_planned_fwd_only_cfg5_pmc.csv")))}
dur = None
for r in csv.DictReader(open(os.path.join(P, "r06_cfg5_profile_only_kernel_stats.csv"))):
    if r["Name"].startswith("void pn_ms3_kernel<0, 0>"):
        dur = float(r["AvgUs"])
gui = pmc["GRBM_GUI_ACTIVE"] / 8.0
ghz = gui / dur / 1e3
mf = pmc["SQ_VALU_MFMA_BUSY_CYCLES"] / 1024.0
gflop = pmc["SQ_VALU_MFMA_BUSY_CYCLES"] / 32.0 * 32768 / 1e9
tf = gflop / dur * 1e3
res = pmc["SQ_VALU_MFMA_BUSY_CYCLES"] / (4.0 * pmc["SQ_WAVE_CYCLES"]) * 2.0
print("""# Clock under load of the dominant kernel (round-5 verdict, item 5) — pn_ms3_kernel<0,0>, the planned mean-shift forward
# launch of 4 shapes inside `bench.py --profile-only` (40 launches; tools/evidence_round6.sh: counters in passes of their own,
# kernel duration from the --kernel-trace --stats pass of the same command: profiles/r06_cfg5_profile_only_kernel_stats.csv;
# this file: tools/clock_record.py from those two files)
""")
print("rocprofv3 average launch duration                     %.2f us" % dur)
print("GRBM_GUI_ACTIVE per launch (sum over the 8 XCDs)      %d  -> %d graphics-clock cycles per XCD" % (pmc["GRBM_GUI_ACTIVE"], gui))
print("  => clock the chip sustains under this kernel        %d / %.2f us = %.3f GHz   (data sheet: 2.4 GHz)" % (gui, dur, ghz))
print("SQ_BUSY_CYCLES per launch (sum over 32 shader engines) %d  -> %d per engine" % (pmc["SQ_BUSY_CYCLES"], pmc["SQ_BUSY_CYCLES"] / 32))
print("SQ_VALU_MFMA_BUSY_CYCLES per launch (sum over SIMDs)  %d  -> %d per SIMD (1 024 SIMDs)" % (pmc["SQ_VALU_MFMA_BUSY_CYCLES"], mf))
print("  => share of the launch's cycles with the matrix pipe busy   %d / %d = %.3f" % (mf, gui, mf / gui))
print("executed work (counter / 32 passes x 32 768 FLOP)     %.1f GFLOP -> %.1f TFLOP/s" % (gflop, tf))
print("  = %.3f of the 2 500 TFLOP/s data-sheet peak (2.4 GHz)" % (tf / 2500.0))
print("  = %.3f of the %d TFLOP/s the matrix cores deliver at the sustained %.2f GHz" % (tf / (2500.0 * ghz / 2.4), 2500.0 * ghz / 2.4, ghz))
print("SQ_WAVE_CYCLES per launch                              %d  -> SQ_VALU_MFMA_BUSY_CYCLES / (4 x SQ_WAVE_CYCLES) = %.3f per wave, x 2 waves per SIMD" % (pmc["SQ_WAVE_CYCLES"], res / 2.0))
print("  => matrix pipe busy %.3f of the cycles in which a SIMD has its two waves resident; %.3f over the whole launch (above):" % (res, mf / gui))
print("     the difference is the launch's ramp and tail (workgroups with list lengths of their own finish at different times)")
print("""
Reading: the round-5 claim ("0.69 of the resident cycles busy, the rest is the power-limited clock") holds in part.  The chip
runs this kernel at %.2f GHz, %.2f of the data-sheet clock (the boxes of the pool differ by a few percent: the evidence runs of the round
measured 1.94 GHz / 477.5 us, 2.00 GHz / 464.9 us and 2.05 GHz / 453.7 us for the same cycle count): that is %.2f of the distance to the peak and not the kernel's.
Of the cycles it has, the matrix pipe is busy %.2f: %.2f are ramp and tail of the planned launch, the remaining %.2f are the
kernel's own — barrier and DMA waits between the two GEMMs of a tile pair and the elementwise stage (exp, row sums) issued
beside the MFMAs.

# sysfs sampler (tools/probes/clock_under_load.py, profiles/r06_clock_sysfs_sampler.txt): the hwmon nodes of this pool are not a
# usable record — on one evidence box freq1_input read 2 393 MHz and power1 504 W in EVERY phase incl. idle; on another box of
# the round (tools/jobs/r6c.sh) 94 MHz / 257 W at idle and a bimodal 94 / 2 390 MHz under the same MFMA load.""" % (
    ghz, ghz / 2.4, 1.0 - ghz / 2.4, mf / gui, res - mf / gui, 1.0 - res))
