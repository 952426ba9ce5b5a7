"""liba52 channel flags (a52dec-0.7.5-cvs/include/a52.h:40-54)."""
A52_CHANNEL = 0
A52_MONO = 1
A52_STEREO = 2
A52_3F = 3
A52_2F1R = 4
A52_3F1R = 5
A52_2F2R = 6
A52_3F2R = 7
A52_CHANNEL1 = 8
A52_CHANNEL2 = 9
A52_DOLBY = 10
A52_CHANNEL_MASK = 15
A52_LFE = 16
A52_ADJUST_LEVEL = 32

NFCHANS = (2, 1, 2, 3, 3, 4, 4, 5, 1, 1, 2)


def out_channels(flags):
    return NFCHANS[flags & A52_CHANNEL_MASK] + (1 if flags & A52_LFE else 0)
