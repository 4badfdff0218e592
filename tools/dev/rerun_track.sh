OUT=gpurun_out/r6_track; mkdir -p $OUT
python3 tools/make_manifest.py --record $OUT
python3 tools/dev/track_time.py 2>&1 | grep -v amdgpu.ids > $OUT/track_time.txt
TRACK_G=1440 TRACK_T=60000 python3 tools/dev/track_time.py 2>&1 | grep -v amdgpu.ids >> $OUT/track_time.txt
cat $OUT/track_time.txt
