#!/bin/bash
# Where the halo-once kernel's time goes: the phase form on dec3's shape with parts of its main loop knocked out
# (robosat_amd/csrc/conv_halo_ko.hip; the knocked-out launches compute garbage, only their time means anything).
# Needs the measurement build of the library: `touch robosat_amd/csrc/conv_igemm_dma.hip && make -C robosat_amd/csrc KO=1`
# before the box visit (the .so travels with the snapshot), and the same without KO=1 afterwards -- the default library has
# neither the knock-out kernels nor the RS_HALO_KO branch.
S="bf16:phase:32,256+64,128,128,128"
for KO in 0 1 2 3 4 0; do
  if [ $KO = 0 ]; then unset RS_HALO_KO; else export RS_HALO_KO=$KO; fi
  echo -n "KO=$KO (0 full, 1 no waits/barriers, 2 no DMA, 3 no fragment reads, 4 no MFMAs): "
  python scripts/bench_layer.py --iters 30 --variants "halo/0" $S 2>/dev/null | tail -1
done
