#!/bin/bash
# F32X3 (six product terms) against the emulated two-plane form (three largest terms only; -DXV2_T0=3 build):
# step time and, with the CPU oracle on, the parity block of each.
for v in libxv2.so libxv2_emu.so libxv2.so libxv2_emu.so; do
  XV2_LIB=/root/repo/xview2_amd/$v python bench.py --no-cpu-baseline --no-encoder-probe --no-other-configs --no-prof --steps 20 --warmup 5 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v', d['value'], d['ms_per_step'], d['loss'])" | tee -a gpurun_out/r04_ab11.log
done
XV2_LIB=/root/repo/xview2_amd/libxv2_emu.so python bench.py --no-encoder-probe --no-other-configs --steps 10 --warmup 3 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('emu parity', json.dumps(d['parity'])); print([(r['kernel'], r['tflops'], r['ms_per_step']) for r in d['roofline']['per_kernel']])" | tee -a gpurun_out/r04_ab11.log
