for mt in 0 4 8; do echo "PTTS_DENSE_BATCHED_MT=$mt"; PTTS_DENSE_BATCHED_MT=$mt python3 tools/conv1d_fft_probe.py 2>/dev/null | grep -E "fft=1 update=1 call 1|wgrad freq=1" | cut -c1-420; done
