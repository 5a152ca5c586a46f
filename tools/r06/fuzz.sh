#!/bin/bash
O=gpurun_out/r06/fuzz; mkdir -p $O
timeout 900 python tools/fuzz_ba.py 240 > $O/ba.txt 2>&1; tail -4 $O/ba.txt
timeout 900 python tools/fuzz_skyline.py > $O/skyline.txt 2>&1; tail -3 $O/skyline.txt
timeout 600 python tools/fuzz_batch.py > $O/batch.txt 2>&1; tail -2 $O/batch.txt
timeout 600 python tools/fuzz_localmap.py > $O/localmap.txt 2>&1; tail -2 $O/localmap.txt
timeout 600 python tools/fuzz_track.py 60 > $O/track.txt 2>&1; tail -2 $O/track.txt
timeout 600 python tools/fuzz_match.py > $O/match.txt 2>&1; tail -2 $O/match.txt
