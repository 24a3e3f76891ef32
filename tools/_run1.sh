echo "938 frames (rows 8-byte aligned)"
timeout 300 python tools/ab.py soundml_amd/lib/libsoundml_amd.so@SMX_INTERLEAVE=0 soundml_amd/lib/libsoundml_amd.so@SMX_INTERLEAVE=2
echo "944 frames (rows 64-byte aligned)"
AB_N=482816 timeout 300 python tools/ab.py soundml_amd/lib/libsoundml_amd.so@SMX_INTERLEAVE=0 soundml_amd/lib/libsoundml_amd.so@SMX_INTERLEAVE=2
echo "960 frames (rows 128-byte aligned)"
AB_N=491008 timeout 300 python tools/ab.py soundml_amd/lib/libsoundml_amd.so@SMX_INTERLEAVE=0 soundml_amd/lib/libsoundml_amd.so@SMX_INTERLEAVE=2
