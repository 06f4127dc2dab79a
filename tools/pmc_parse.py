"""Summarise rocprofv3 --pmc passes (one sub-directory per pass, rocpd sqlite output): mean counter value per launch
and mean kernel duration for the GEMM kernels.  usage: pmc_parse.py <dir>"""
import sqlite3,glob,sys,json
res={}
for d in sorted(glob.glob(sys.argv[1]+"/*/")):
    fs=glob.glob(d+"*.db")
    if not fs: continue
    c=sqlite3.connect(fs[0])
    tabs=[r[0] for r in c.execute("select name from sqlite_master where type='table'")]
    pc=[t for t in tabs if 'pmc_event' in t]; ks=[t for t in tabs if 'kernel_symbol' in t][0]; kd=[t for t in tabs if 'kernel_dispatch' in t][0]
    pi=[t for t in tabs if 'info_pmc' in t]
    if not pc: continue
    q=f"select s.kernel_name, i.name, avg(e.value), count(*) from {pc[0]} e join {kd} d on e.event_id=d.event_id join {ks} s on d.kernel_id=s.id join {pi[0]} i on e.pmc_id=i.id group by s.kernel_name, i.name"
    try:
        for kn, dur, n in c.execute(f"select s.kernel_name, avg(d.end-d.start), count(*) from {kd} d join {ks} s on d.kernel_id=s.id group by s.kernel_name"):
            if "gemm" in kn or "Cijk" in kn or "attn" in kn:
                res.setdefault(kn[:60],{}).setdefault("_avg_us_under_pmc", []).append(round(dur/1e3,1))
        for kn,cn,v,n in c.execute(q):
            if "gemm" in kn or "Cijk" in kn or "attn" in kn:
                res.setdefault(kn[:60],{})[cn]=v
    except Exception as ex:
        print("ERR",d,ex)
print(json.dumps(res,indent=1))
