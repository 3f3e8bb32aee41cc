import sys, statistics
rows=[l.strip().split(',') for l in open(sys.argv[1]) if l.strip()]
inf=[float(r[1]) for r in rows if r[0]=="performance/step_inference_fps"]
tot=[float(r[1]) for r in rows if r[0]=="performance/step_inference_rl_update_fps"]
rw=[float(r[1]) for r in rows if r[0]=="rewards/iter"]
print("rollout fps median (all / last 50): %.1f M / %.1f M   total fps: %.1f M / %.1f M   reward last %.1f" % (statistics.median(inf)/1e6, statistics.median(inf[-50:])/1e6, statistics.median(tot)/1e6, statistics.median(tot[-50:])/1e6, rw[-1]))
