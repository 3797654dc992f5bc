# 30-run evaluation (20,000 steps) against KBRL_ROUNDS / KBRL_EVAL_CHUNK; the first run of a box is slower: one dummy run first
STEPS=3000 RUNS=30 PROFILE=tdl timeout 600 bash tools/eval_scale.sh > /dev/null
for CFG in ${CFGS:-"3 64" "4 64" "5 64" "6 64" "3 64" "4 64" "5 64"}; do set -- $CFG
KBRL_ROUNDS=$1 KBRL_EVAL_CHUNK=$2 STEPS=${STEPS:-20000} RUNS=30 PROFILE=tdl timeout 600 bash tools/eval_scale.sh | head -1 | cut -c1-110 | sed "s/^/ROUNDS=$1 CHUNK=$2: /"
done
