#!/bin/bash
# Same-box A/B against the round-2 tree (a git worktree at de50c57 built into ./_r02, git-ignored): runs ON THE GPU BOX.
# Order alternates (r02, cur, cur, r02) so that a clock / thermal drift inside the call does not favour one side.
one() {
  python tools/gemm_bench.py -1,4 2>&1 | grep -E "qkv|out|ffn|sq4096"
  python bench.py --no-cpu-baseline --no-api 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['roofline']['per_launch_tflops']); print(d['kernel_ms_per_forward'])"
}
echo "== r02 tree (1)"; (cd _r02 && one)
echo "== current tree (1)"; one
echo "== current tree (2)"; one
echo "== r02 tree (2)"; (cd _r02 && one)
