#!/bin/bash
# Same-box comparison of the round-4 tree (git archive of the round-4 commit, unpacked to .r04_tree/ by the caller: not tracked) against HEAD:
# builds the old library in its own directory, then runs the two bench.py alternately (A B A B A B) on the one GPU.
set -e
test -d .r04_tree || { echo "unpack the round-4 commit first: mkdir .r04_tree && git archive <commit> | tar -x -C .r04_tree"; exit 1; }
(cd .r04_tree/soundeventdetection-pytorch_amd/csrc && make -j14 > /tmp/mk_r04.log 2>&1) || { tail -5 /tmp/mk_r04.log; exit 1; }
for i in 1 2 3; do
  (cd .r04_tree && python bench.py --steps 40 --warmup 8 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('round 4 tree : %.4f ms/step  %.1f clips/s   dominant %s %.4f ms' % (d['ms_per_step'], d['value'], d['roofline']['kernel'][:34], d['roofline']['avg_ms']))")
  python bench.py --steps 40 --warmup 8 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('HEAD         : %.4f ms/step  %.1f clips/s   dominant %s %.4f ms' % (d['ms_per_step'], d['value'], d['roofline']['kernel'][:34], d['roofline']['avg_ms']))"
done
