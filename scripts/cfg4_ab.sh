set -x
C4="--encoder resnest101 --type post --dmg_model siamese --loss_str focal+dice --no-cpu-baseline --no-encoder-probe --no-other-configs --no-prof --steps 10 --warmup 4"
python bench.py $C4 2>&1 | tail -1 | cut -c1-400
XV2_BN_FOLD=0 python bench.py $C4 2>&1 | tail -1 | cut -c1-400
XV2_BN_ROWS=0 python bench.py $C4 2>&1 | tail -1 | cut -c1-400
python bench.py $C4 --precision 16 2>&1 | tail -1 | cut -c1-400
