for nf in 1 4 16; do for st in 2 4 6 8; do echo "nf=$nf streams=$st: $(ATLAS_AMD_FFT_STREAMS=$st timeout 300 python tools/bench_grid.py O1280 1279 $nf 2>&1 | tail -1)"; done; done
