#!/bin/bash
# One workload of bench.py through several builds of the library, interleaved, three rounds: prints ms_per_step per build and round.
#   gpurun -- 'bash scripts/ab_bench_variants.sh cfg4_lens_f32 ray-tracing-ultrasound_amd/librtus.so variants/librtus_lens6.so ...'
WL=$1; shift
for round in 1 2 3; do
  for lib in "$@"; do
    RTUS_LIB=$lib python3 bench.py --workload $WL --steps 60 --warmup 10 --no-extra --no-cpu-baseline 2>/dev/null | python3 -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('round $round  %-40s ms_per_step %.5f  value %.1f' % ('$lib', d['ms_per_step'], d['value']))"
  done
done
