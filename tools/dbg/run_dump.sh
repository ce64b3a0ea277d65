for i in 0 24 25 26; do echo impl $i; python tools/dbg/gemm8_dump.py $i 2>&1 | grep -E "bad count|^\(" | head -4; done
