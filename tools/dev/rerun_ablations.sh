OUT=gpurun_out/r6_abl; mkdir -p $OUT
python3 tools/make_manifest.py --record gpurun_out/r6_abl
bash tools/dev/ab_lib.sh tools/_variants/libmicloc_hip_ws_k4.so "--steps 40 --warmup 4" > $OUT/ablation_kstep.txt 2>&1
bash tools/dev/ab_lib.sh tools/_variants/libmicloc_hip_stht_valu.so "--steps 40 --warmup 4" > $OUT/ablation_stht.txt 2>&1
bash tools/dev/ab_lib.sh tools/_variants/libmicloc_hip_stht_one_tile.so "--steps 40 --warmup 4" > $OUT/ablation_stht_walk.txt 2>&1
bash tools/dev/ab_bf.sh default tools/_variants/libmicloc_hip_ws_sparse_lif.so > $OUT/ablation_sparse_lif.txt 2>&1
grep -v amdgpu $OUT/ablation_kstep.txt $OUT/ablation_stht.txt $OUT/ablation_stht_walk.txt $OUT/ablation_sparse_lif.txt
