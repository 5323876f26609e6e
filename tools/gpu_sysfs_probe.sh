#!/bin/bash
# which hwmon belongs to the GPU HIP exposes (bench.py GpuSampler)
for hw in /sys/class/drm/card*/device/hwmon/hwmon*; do
  echo "$hw $(grep PCI_SLOT_NAME $hw/../../uevent) power=$(cat $hw/power1_average 2>/dev/null || cat $hw/power1_input 2>/dev/null) freq=$(cat $hw/freq1_input 2>/dev/null) $(ls $hw | tr '\n' ' ')"
done
python - <<'PY'
import torch
p = torch.cuda.get_device_properties(0)
print('torch device 0:', p.name, [ (a, getattr(p, a)) for a in dir(p) if 'pci' in a ])
PY
rocm-smi --showpower --showclocks --json 2>/dev/null | head -c 1500
