cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
rm -f gpurun_out/sweep.log
for n in 1024 4096 8192 16384 32768 131072; do
  timeout 300 python bench.py --cpu-seconds 0 --ppo-iters 0 --steps 1000 --warmup 100 --check-steps 0 --envs $n 2>/dev/null | grep metric | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('envs=$n', round(d['value']/1e6,2), 'M env-steps/s', round(d['roofline']['avg_launch_us'],2), 'us', 'fp32_frac', round(d['roofline_fp32']['frac'],4))" >> gpurun_out/sweep.log
done
python - <<'PY' >> gpurun_out/sweep.log 2>&1
# PCIe-inclusive rate of the numpy (reference-style) boundary
import sys, time, os
sys.path.insert(0, os.getcwd())
import numpy as np, yaml, torch
import high_speed_quadrupedal_locomotion_by_irrl_amd as pkg
from high_speed_quadrupedal_locomotion_by_irrl_amd.flexible_robot import FlexibleGymEnv
from high_speed_quadrupedal_locomotion_by_irrl_amd.vec_env import RaisimGymVecEnv
cfg = yaml.safe_load(open(os.path.join(pkg.__BLACKPANTHER_V55_RESOURCE_DIRECTORY__, "bp5_imitation.yaml")))["environment"]
env = RaisimGymVecEnv(FlexibleGymEnv(pkg.__BLACKPANTHER_V55_RESOURCE_DIRECTORY__, yaml.safe_dump(cfg)))
a = np.clip(0.3 * np.random.RandomState(0).normal(size=(4096, 12)), -1, 1).astype(np.float32)
for _ in range(50): env.step(a)
t = time.perf_counter()
for _ in range(1000): env.step(a)
dt = time.perf_counter() - t
print("host numpy path (RaisimGymVecEnv.step, H2D+kernel+D2H+copies): %.1f us/step, %.2f M env-steps/s" % (dt * 1e3, 4096 * 1000 / dt / 1e6))
PY
echo done
