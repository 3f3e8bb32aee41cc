cd gpurun_out
for cfg in "True 42" "True 7" "False 7"; do
  set -- $cfg
  rm -rf runs
  python3 ../train.py task=Vine5LinkMovingBase num_envs=16384 max_iterations=2000 headless=True experiment=soak seed=$2 train.params.config.env_stats_every=0 train.params.config.save_frequency=0 train.params.config.mixed_precision=$1 > soak_$1_$2.log 2>&1
  python3 - <<PY
import csv, math
rows=[r for r in csv.reader(open("runs/soak/summaries/scalars.csv"))]
rew=[(int(r[2]), float(r[1])) for r in rows if r[0]=="rewards/iter"]
low=[(i,round(v)) for i,v in rew if i>100 and v<900]
print("mixed=$1 seed=$2 last", round(rew[-1][1]), "min after 100:", round(min(v for i,v in rew if i>100)), "iters<900:", len(low), "mean after 100:", round(sum(v for i,v in rew if i>100)/sum(1 for i,v in rew if i>100),1), "first lows", low[:6])
PY
done
rm -rf runs
