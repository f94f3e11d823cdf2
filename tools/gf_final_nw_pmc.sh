# FETCH_SIZE of k_gf_ws_final with one and four waves per strip; GPU box:  bash tools/gf_final_nw_pmc.sh
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for n in 1 4; do
  export UWIP_GF_FINAL_NW=$n
  timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE -d gpurun_out/fnw_pmc_$n --output-format csv -- python3 tools/kernel_times.py 64 2 > gpurun_out/fnw_pmc_$n.log 2>&1 || exit 1
done
python3 - <<'PY'
import csv,glob,collections
for n in (1,4):
    f=glob.glob(f'gpurun_out/fnw_pmc_{n}/**/*counter_collection.csv',recursive=True)[0]
    acc=collections.defaultdict(lambda:[0,0.0])
    for r in csv.DictReader(open(f)):
        k=r['Kernel_Name']
        if 'k_gf_ws' in k:
            key=k.split('(')[0][-60:]
            acc[key][0]+=1; acc[key][1]+=float(r['Counter_Value'])
    for k,(c,v) in acc.items(): print(n,k,c,'KB/launch',round(v/c))
PY
