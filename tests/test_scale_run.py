"""tools/scale_run.sh -- the script the first run on an 8-GPU node goes through -- rehearsed on the CPU at
N = 8: every launch path of it (bench-style rank jobs at 1 / 2 / 4 / 8 ranks under torch.distributed.run over gloo, the
`main.py --gpus 8`-style shard processes with a remainder and the host-side concat) and the one table it prints.  The forward is
the CPU oracle double (tests/rehearse_scale_rank.py); rank counting, sharding, concat and the gradient exchange are the product's."""
import os
import re
import subprocess

from conftest import ROOT


def test_scale_run_rehearsal_at_eight_ranks(tmp_path):
    out = str(tmp_path / "scale")
    r = subprocess.run(["bash", os.path.join(ROOT, "tools", "scale_run.sh"), "--rehearse-cpu", "8", "--out", out],
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-1500:])
    rows = re.findall(r"^(infer|train)\s+(\d+)\s+(\d+) (?:candidate-variants|sites)/s\s+([0-9.]+)\s+([0-9.]+)\s+(\d+)\s+(.*)$", r.stdout, re.M)
    got = {(m, int(n)): (int(seen), tail) for m, n, _, _, _, seen, tail in rows}
    for mode in ("infer", "train"):
        for n in (1, 2, 4, 8):
            assert (mode, n) in got, (mode, n, r.stdout[-2000:])
            assert got[(mode, n)][0] == n, "ranks_seen must equal the ranks launched"
    assert got[("infer", 8)][1].startswith("no collective on the data path")
    assert got[("train", 1)][1].startswith("no exchange")
    assert re.match(r"[0-9.]+ ms exposed \+ [0-9.]+ ms normalisers, all-reduce", got[("train", 8)][1])
    shards = re.findall(r"shard (\d)/8 on device cpu: (\d+) sites", r.stdout)
    assert len(shards) == 8 and sum(int(s) for _, s in shards) == 323 and {int(s) for _, s in shards} == {40, 41}
    assert re.search(r"whole job: 323 sites in [0-9.]+ s = \d+ sites/s; host-side concat [0-9.]+ s", r.stdout)
    assert os.path.isfile(os.path.join(out, "scale_run.log"))
