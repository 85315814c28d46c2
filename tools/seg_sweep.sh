#!/bin/bash
# segment length of the overlap-save frequency-domain Conv1D (GPU box): forward, transforms, weight gradient per setting
for s in 80 100 200 0; do echo "PTTS_CONV1D_FFT_SEG=$s"; PTTS_CONV1D_FFT_SEG=$s python3 tools/conv1d_fft_probe.py 2>/dev/null | grep -E "fft=1 update=1 call 1|fft=1 update=0 call 2|wgrad freq=1|max error" | cut -c1-400; done
