"""Grid rounds per kernel from a rocprofv3 kernel trace (k_kernel_trace.csv): workgroups, waves per workgroup, VGPRs and LDS give
the workgroups a CU can hold; a grid slightly above a multiple of (CUs x resident) pays a whole extra round for a few workgroups
(dcn_bwd_weight_mfma: 774 workgroups, one resident per CU, 4 rounds for 3.02 — found by hand, hence this tool).
    python tools/kt_rounds.py <k_kernel_trace.csv> [n_cus=256]"""
import csv, sys, collections
path = sys.argv[1]; ncu = int(sys.argv[2]) if len(sys.argv) > 2 else 256
rows = collections.OrderedDict()
for r in csv.DictReader(open(path)):
    if r["Kind"] != "KERNEL_DISPATCH": continue
    wg = int(r["Workgroup_Size_X"]) * int(r["Workgroup_Size_Y"]) * int(r["Workgroup_Size_Z"])
    grid = int(r["Grid_Size_X"]) * int(r["Grid_Size_Y"]) * int(r["Grid_Size_Z"])
    # rocprofv3 reports the register count of gfx950 kernels HALVED (dcn_bwd_data_own: 251 VGPRs in the ISA, 128 in the trace) and
    # LDS_Block_Size without the dynamic part (0 for kernels that take all of theirs at launch): resident counts of such kernels
    # are upper bounds
    key = (r["Kernel_Name"][:110], grid // max(wg, 1), wg, 2 * (int(r["VGPR_Count"]) + int(r["Accum_VGPR_Count"])), int(r["LDS_Block_Size"]))
    d = rows.setdefault(key, [0, 0.0])
    d[0] += 1; d[1] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
print(f"{'kernel':<72} {'WGs':>7} {'thr':>4} {'vgpr':>4} {'lds':>6} {'res/CU':>6} {'rounds':>7} {'us':>8} {'calls':>5}")
for (name, nwg, wg, vgpr, lds), (calls, us) in rows.items():
    if us / calls < 20: continue
    waves = (wg + 63) // 64
    per_simd = max(1, min(8, 512 // max(vgpr, 1)))       # waves per SIMD by registers (512 unified registers per lane)
    by_reg = (per_simd * 4) // waves if waves <= per_simd * 4 else 0
    by_lds = (160 * 1024) // lds if lds else 99
    by_waves = 32 // waves                                # (at most 32 waves per CU counted here)
    res = max(1, min(by_reg, by_lds, by_waves))
    rounds = nwg / (ncu * res)
    frac = rounds - int(rounds)
    flag = " <-- tail" if (rounds < 6 and 0 < frac < 0.25 and rounds > 1) else ""
    print(f"{name[:72]:<72} {nwg:>7} {wg:>4} {vgpr:>4} {lds:>6} {res:>6} {rounds:>7.2f} {us / calls:>8.1f} {calls:>5}{flag}")
