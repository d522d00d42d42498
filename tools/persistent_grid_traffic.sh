REPO=$PWD; export TMPDIR=/tmp
cd /tmp
for g in base 512 1024; do
  if [ $g = base ]; then E=""; else E="MSDFHIP_PERSISTENT_ROUNDS=1 MSDFHIP_PERSISTENT_GRID=$g"; fi
  env $E rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $REPO/gpurun_out/pgt_w_$g -o w -- python $REPO/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-extras > $REPO/gpurun_out/pgt_w_$g.log 2>&1
  env $E rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $REPO/gpurun_out/pgt_f_$g -o f -- python $REPO/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-extras > $REPO/gpurun_out/pgt_f_$g.log 2>&1
done
cd $REPO
python - <<PY
import sqlite3,glob
for g in ("base","512","1024"):
    for kind,ctr in (("w","WRITE_SIZE"),("f","FETCH_SIZE")):
        db=glob.glob("gpurun_out/pgt_%s_%s/**/*.db"%(kind,g),recursive=True)[0]
        cur=sqlite3.connect(db).cursor()
        rows=cur.execute("select kernel_name,sum(value),count(*) from counters_collection where counter_name=? group by kernel_name",(ctr,)).fetchall()
        steps=max(n for k,_,n in rows if "k_ec_fast" in k)
        tot=sum(v for k,v,_ in rows if "k_distance" in k)/steps
        print(g,ctr,"KiB per step in k_distance:",round(tot), {k.split("(")[0][-28:]:round(v/steps) for k,v,_ in rows if "k_distance" in k})
PY
find gpurun_out -name "*.db" -path "*pgt_*" -delete
