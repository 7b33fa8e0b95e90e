R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
bash scripts/measure_round3.sh
O=$R/gpurun_out
for i in 1 2; do
SLIDERS_OVERLAP_FROZEN=1 timeout 300 python bench.py --no-cpu-baseline --no-extra --no-roofline --steps 10 > $O/r03_overlap_on_$i.json 2>/dev/null; python -c "import json;r=json.load(open('$O/r03_overlap_on_$i.json'));print('overlap on ',r['value'],r['ms_per_step'])"
timeout 300 python bench.py --no-cpu-baseline --no-extra --no-roofline --steps 10 > $O/r03_overlap_off_$i.json 2>/dev/null; python -c "import json;r=json.load(open('$O/r03_overlap_off_$i.json'));print('overlap off',r['value'],r['ms_per_step'])"
done
