for V in 0 256 128 0 256; do echo "== ITG_NT_BPIX=$V"; ITG_NT_BPIX=$V python tools/conv_bench.py "G b" 2>/dev/null | grep -v "^sum"; done
