# measurement: the Dense weight-gradient kernel with and without its atomic flush (run on the GPU box)
for v in "X=1" "PTTS_DENSE_WGRAD_NOFLUSH=1"; do echo "== $v"; env $v python tools/dense_split_probe.py 2>&1 | grep "^Kin" | head -3; done
