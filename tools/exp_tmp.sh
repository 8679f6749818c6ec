python3 -m pytest tests/test_gpu_tokens.py tests/test_gpu_configs.py -x -q 2>&1 | tail -3
for wh in "1024 1024 256" "1920 1080 256" "2048 2048 256" "4096 4096 128"; do set -- $wh; python3 tools/enc_stages.py $1 $2 $3 5 2>&1 | grep -o "encode of.*\|'k_tok': [0-9.]*\|'k_emit_tok': [0-9.]*"; done
