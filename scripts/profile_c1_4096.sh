export TMPDIR=/tmp
bash scripts/profile_ppo_graphed.sh r04_c1_4096 --num-envs 4096 --obs-type TIP_AND_CART_AND_OBJ_INFO --no-other-configs --override task.env.CREATE_SHELF=False --override task.env.maxEpisodeLength=100 --override task.env.SUCCESS_DIST=0.04 --override task.env.MIN_TARGET_Y=-0.4 --override task.env.MAX_TARGET_Y=0.4 --override task.env.MIN_TARGET_Z=0.55 --override task.env.MAX_TARGET_Z=0.7 --override RAIL_SOFT_LIMIT=0.25 --override RAIL_P_GAIN=30 --override RAIL_ACCELERATION=6 > /dev/null 2>&1
head -30 gpurun_out/prof_ppo_r04_c1_4096/summary.txt | cut -c1-150
tail -1 gpurun_out/prof_ppo_r04_c1_4096/bench.log | cut -c1-300
