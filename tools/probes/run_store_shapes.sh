cd /root/repo/tools/probes
for mode in 0 1 2; do for al in 0 64; do ./store_shape_probe 16 938 $mode 0 $al; done; done
for mode in 0 1 2; do for al in 0 64 128; do ./store_shape_probe 32 938 $mode 0 $al; done; done
./store_shape_probe 64 938 2 0 0; ./store_shape_probe 64 938 2 0 128; ./store_shape_probe 64 938 1 0 128
./store_shape_probe 16 944 2 0 0; ./store_shape_probe 32 960 2 0 0
