#!/bin/bash
# Round-3 randomised campaign: gpurun -- 'bash profiles/run_fuzz_r03.sh'  ->  gpurun_out/fuzz_r03.txt (copy to profiles/r03_fuzz_campaign.txt)
# New seeds; the decode scripts under the split front end (modes 4 and 5: fuzz shapes are small, so the default would be the
# fused kernel) beside the default choice, the encoder under both packers.
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/fuzz_r03.txt
mkdir -p $R/gpurun_out; : > $OUT
t() { name=$1; shift; "$@" > /tmp/fz_$name.log 2>&1; echo "$name rc=$? $(tail -1 /tmp/fz_$name.log)" | tee -a $OUT; }
cd $R
AC3MI_DECODE_MODE=4 t decode_m4 python tests/fuzz_decode.py 150 3001
AC3MI_DECODE_MODE=5 t decode_m5 python tests/fuzz_decode.py 100 3002
t decode_auto python tests/fuzz_decode.py 100 3003
AC3MI_DECODE_MODE=4 t pcm_m4 python tests/fuzz_pcm.py 400 3004
AC3MI_DECODE_MODE=5 t pcm_m5 python tests/fuzz_pcm.py 200 3005
t pcm_auto python tests/fuzz_pcm.py 200 3006
AC3MI_DECODE_MODE=4 t mix_m4 python tests/fuzz_mixlevel.py 150 3007
AC3MI_DECODE_MODE=4 t corrupt_m4 python tests/fuzz_corrupt.py 100 3008
AC3MI_DECODE_MODE=5 t corrupt_m5 python tests/fuzz_corrupt.py 60 3009
AC3MI_DECODE_MODE=4 t corrupt_st_m4 python tests/fuzz_corrupt.py 100 3010 2
AC3MI_ENCODE_MODE=1 t encode_m1 python tests/fuzz_encode.py 400 3011
AC3MI_ENCODE_MODE=2 t encode_m2 python tests/fuzz_encode.py 400 3012
t stream_auto python tests/fuzz_stream.py 200 3013
AC3MI_DECODE_MODE=4 AC3MI_ENCODE_MODE=2 t stream_m4 python tests/fuzz_stream.py 100 3014
t transcode_auto python tests/fuzz_transcode.py 100 3015
AC3MI_DECODE_MODE=4 AC3MI_ENCODE_MODE=1 t transcode_m4 python tests/fuzz_transcode.py 100 3016
