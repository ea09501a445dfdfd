hipcc --offload-arch=gfx950 -O3 -o /tmp/mfma_coissue tools/dev/mfma_coissue.hip 2>&1 | grep -i "error" ; timeout -k 10 120 /tmp/mfma_coissue
