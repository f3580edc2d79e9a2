#!/bin/bash
# round 4, sixth GPU call: same-box A/B of "the final stage only observes" (base) against the build before it (prev), then the profiles
cd "$(dirname "$0")/.."
bash tools/gpu_ab.sh r4g/ab base prev
bash tools/gpu_profiles.sh 2 r4
bash tools/gpu_profiles.sh 3 r4
