#!/bin/bash
# Round-3 randomised campaign: gpurun -- 'bash profiles/run_fuzz_r03.sh'  ->  gpurun_out/fuzz_r03.txt (copy to profiles/r03_fuzz_campaign.txt)
# New seeds; the decode scripts under the split front end (modes 4 and 5: fuzz shapes are small, so the default would be the
# fused kernel) beside the default choice, the encoder under both packers.
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/fuzz_r03.txt
mkdir -p $R/gpurun_out; : > $OUT
t() { name=$1; shift; "$@" > /tmp/fz_$name.log 2>&1; echo "$name rc=$? $(tail -1 /tmp/fz_$name.log)" | tee -a $OUT; }
cd $R
AC3MI_DECODE_MODE=4 t decode_m4 python tests/fuzz_decode.py 150 3101
AC3MI_DECODE_MODE=5 t decode_m5 python tests/fuzz_decode.py 100 3102
t decode_auto python tests/fuzz_decode.py 100 3103
AC3MI_DECODE_MODE=4 t pcm_m4 python tests/fuzz_pcm.py 400 3104
AC3MI_DECODE_MODE=5 t pcm_m5 python tests/fuzz_pcm.py 200 3105
t pcm_auto python tests/fuzz_pcm.py 200 3106
AC3MI_DECODE_MODE=4 t mix_m4 python tests/fuzz_mixlevel.py 150 3107
AC3MI_DECODE_MODE=4 t corrupt_m4 python tests/fuzz_corrupt.py 100 3108
AC3MI_DECODE_MODE=5 t corrupt_m5 python tests/fuzz_corrupt.py 60 3109
AC3MI_DECODE_MODE=4 t corrupt_st_m4 python tests/fuzz_corrupt.py 100 3110 2
AC3MI_ENCODE_MODE=1 t encode_m1 python tests/fuzz_encode.py 400 3111
AC3MI_ENCODE_MODE=2 t encode_m2 python tests/fuzz_encode.py 400 3112
t stream_auto python tests/fuzz_stream.py 200 3113
AC3MI_DECODE_MODE=4 AC3MI_ENCODE_MODE=2 t stream_m4 python tests/fuzz_stream.py 100 3114
t transcode_auto python tests/fuzz_transcode.py 100 3115
AC3MI_DECODE_MODE=4 AC3MI_ENCODE_MODE=1 t transcode_m4 python tests/fuzz_transcode.py 100 3116
