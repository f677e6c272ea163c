#!/bin/bash
# The reference's whole sweep list (go-run-encoder-rtf.single-gpu-3x3-g5.sh:59-103) on one MI355X: tools/rtf_sweep.py per model, then
# tools/rtf_tables.py over all of them.  tools/rtf_sweep_all_models.sh <out_dir> [rwkv|mamba|tables]   (two GPU calls: the Mamba-2
# models in the YAML's fp32 take minutes each)
set -e
O=$1; W=${2:-rwkv}; mkdir -p $O
S="python3 tools/rtf_sweep.py --passes 3"
if [ $W = rwkv ]; then
  $S --out $O/rwkv_bi_12L > $O/rwkv_bi_12L.log 2>&1; echo "bi 12" >> $O/progress.log
  $S --direction uni --out $O/rwkv_uni_12L > $O/rwkv_uni_12L.log 2>&1
  $S --direction uni --num-blocks 18 --out $O/rwkv_uni_18L > $O/rwkv_uni_18L.log 2>&1; echo "uni" >> $O/progress.log
  for n in 18 24 30; do $S --num-blocks $n --out $O/rwkv_bi_${n}L > $O/rwkv_bi_${n}L.log 2>&1; echo "bi $n" >> $O/progress.log; done
  for v in "only:-1" "bi11:11" "bi9-11:9,10,11" "BiFirst:0" "BiLast6:6,7,8,9,10,11"; do
    n=${v%%:*}; l=${v##*:}
    $S --slot dir_drop_both --dir-dropout-layers=$l --name rwkvbi_12L_alt-$n-GPU --out $O/rwkvbi_12L_alt-$n > $O/alt-$n.log 2>&1
    $S --slot dir_drop_both --dir-dropout-layers=$l --alt-decoding --name rwkvbi_12L_alt-${n}_altdec-GPU --out $O/rwkvbi_12L_alt-${n}_altdec > $O/alt-${n}_altdec.log 2>&1
    echo "alt $n" >> $O/progress.log
  done
elif [ $W = mamba ]; then
  $S --slot mamba_att --out $O/mamba2bi_12L > $O/mamba2bi_12L.log 2>&1; echo "mamba bi" >> $O/progress.log
  $S --slot mamba_att --direction uni --out $O/mamba2_uni_12L > $O/mamba2_uni_12L.log 2>&1; echo "mamba uni" >> $O/progress.log
fi
if ls $O/*.jsonl > /dev/null 2>&1; then python3 tools/rtf_tables.py $O/rtf_tables_all_models.md $O/*.jsonl > /dev/null; fi
