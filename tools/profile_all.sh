#!/bin/bash
# full r03 profile set
for S in cbox bunny scene1; do
  bash tools/profile_gpu.sh r03_$S --scene $S > gpurun_out/prof_r03_$S.log 2>&1 || echo "profile $S failed"
  echo "done $S"
done
bash tools/profile_gpu.sh r03_buddha_standin --scene buddha_standin --steps 6 > gpurun_out/prof_r03_buddha_standin.log 2>&1 || echo "profile buddha failed"
echo "done buddha"
bash tools/profile_gpu.sh r03_dragon_standin --scene dragon_standin --steps 4 --warmup 1 > gpurun_out/prof_r03_dragon_standin.log 2>&1 || echo "profile dragon failed"
echo "done dragon"
