import csv,glob,sys,statistics as st
k=glob.glob((sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/gap_prof")+"/*/*kernel_trace.csv")[0]
ops=[]
for r in csv.DictReader(open(k)): ops.append((int(r["Start_Timestamp"]),int(r["End_Timestamp"]),r["Kernel_Name"].split("(")[0].replace("odo::","")))
ops.sort()
lm=[o for o in ops if o[2] in ("lm_coarse_kernel","lm_fine_kernel")]
gaps_cf=[];gaps_fc=[]
for a,b in zip(lm[:-1],lm[1:]):
    g=(b[0]-a[1])/1e3
    if a[2]=="lm_coarse_kernel" and b[2]=="lm_fine_kernel": gaps_cf.append(g)
    if a[2]=="lm_fine_kernel" and b[2]=="lm_coarse_kernel": gaps_fc.append(g)
gaps_fc=[g for g in gaps_fc if g<200]
print("coarse->fine gap us: median %.1f mean %.1f"%(st.median(gaps_cf), st.mean(gaps_cf)))
print("fine->next coarse gap us: median %.1f mean %.1f p90 %.1f n %d"%(st.median(gaps_fc), st.mean(gaps_fc), sorted(gaps_fc)[int(len(gaps_fc)*0.9)], len(gaps_fc)))
co=[(o[1]-o[0])/1e3 for o in lm if o[2]=="lm_coarse_kernel"]; fi=[(o[1]-o[0])/1e3 for o in lm if o[2]=="lm_fine_kernel"]
print("coarse mean %.1f median %.1f; fine mean %.1f median %.1f"%(st.mean(co), st.median(co), st.mean(fi), st.median(fi)))
i0=[i for i,o in enumerate(ops) if o[2]=="lm_coarse_kernel"][50]
t0=ops[i0][0]
for s,e,n in ops[i0:i0+26]: print("%8.1f -> %8.1f (+%6.1f) %s"%((s-t0)/1e3,(e-t0)/1e3,(e-s)/1e3,n))
