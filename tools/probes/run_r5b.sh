#!/bin/bash
# round 5: the step with and without the stored keep bits (interleaved, product library), both bench shapes
mkdir -p gpurun_out
bash tools/probes/ab.sh gpurun_out/r5b_ab512 3 "LIBV=product SET:attn_keep_bits=1" "LIBV=product SET:attn_keep_bits=0" > gpurun_out/r5b_ab512.txt 2>&1
EXTRA="--frames 1024 --phonemes 128" bash tools/probes/ab.sh gpurun_out/r5b_ab1024 3 "LIBV=product SET:attn_keep_bits=1" "LIBV=product SET:attn_keep_bits=0" > gpurun_out/r5b_ab1024.txt 2>&1
cat gpurun_out/r5b_ab512.txt gpurun_out/r5b_ab1024.txt
