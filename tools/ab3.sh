# usage: bash tools/ab3.sh libA libB [libC]: pairwise interleaved A/B of library variants on one box
set -e
A=$1; B=$2; C=$3
python tools/ab.py $A $B 3
if [ -n "$C" ]; then python tools/ab.py $B $C 3; fi
