"""bench.py's host-side logic that needs no GPU: the self-launch of `--gpus N`, the one-line device-count error, the
walked-position count behind the gather leg's bytes (a restatement of the kernel's rule, slice by slice)."""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def _args(argv):
    old = sys.argv
    sys.argv = ['bench.py'] + argv
    try:
        return bench.parse()
    finally:
        sys.argv = old


def test_bare_multi_gpu_command_relaunches_itself_under_torch_distributed_run(monkeypatch):
    """`python bench.py --gpus 2 ...` with no launcher: the process replaces itself by the N-rank job (one rank per GPU,
    static rendezvous on 127.0.0.1), passing its own arguments through."""
    seen = {}
    monkeypatch.delenv('WORLD_SIZE', raising=False)
    monkeypatch.delenv('RANK', raising=False)
    monkeypatch.setenv('R4R_DIST_BACKEND', 'gloo')           # (no GPU here: the device-count check is RCCL's)
    monkeypatch.setattr(os, 'execv', lambda exe, cmd: seen.update(exe=exe, cmd=cmd))
    monkeypatch.setattr(sys, 'argv', ['bench.py', '--gpus', '2', '--steps', '20', '--warmup', '5'])
    bench.self_launch(bench.parse())
    cmd = seen['cmd']
    assert cmd[:3] == [sys.executable, '-m', 'torch.distributed.run']
    assert '--nproc-per-node' in cmd and cmd[cmd.index('--nproc-per-node') + 1] == '2'
    assert cmd[cmd.index('--master-addr') + 1] == '127.0.0.1' and int(cmd[cmd.index('--master-port') + 1]) > 0
    i = cmd.index(os.path.join(ROOT, 'bench.py'))
    assert cmd[i + 1:] == ['--gpus', '2', '--steps', '20', '--warmup', '5']


def test_no_relaunch_inside_a_launched_job_or_at_one_gpu(monkeypatch):
    monkeypatch.setattr(os, 'execv', lambda *a: pytest.fail('re-launched'))
    monkeypatch.setenv('WORLD_SIZE', '2')
    bench.self_launch(_args(['--gpus', '2']))
    monkeypatch.delenv('WORLD_SIZE')
    monkeypatch.delenv('RANK', raising=False)
    bench.self_launch(_args(['--gpus', '1']))


def test_more_ranks_than_gpus_is_one_clear_line(monkeypatch):
    monkeypatch.delenv('R4R_DIST_BACKEND', raising=False)
    with pytest.raises(SystemExit) as e:
        bench.check_device_count(64)
    assert '--gpus 64' in str(e.value) and 'visible' in str(e.value) and '\n' not in str(e.value)


def _walked_by_the_kernels_rule(doc):
    """proj_gather_max_kernel, one document: P = T + 2 positions in segments of 128, a segment dealt to four workers in
    equal slices (32 positions of a full segment, ceil(len / 4) of a shorter one); a worker's tokens are p_lo - 2 ..
    p_hi - 1; a token outside [0, T) has slot -1; all slots equal -> one position walked."""
    T = len(doc)
    P = T + 2
    walked = 0
    for seg in range((P + 127) // 128):
        seg_len = min(128, P - seg * 128)
        slen = 32 if seg_len == 128 else (seg_len + 3) // 4
        for worker in range(4):
            p_lo = seg * 128 + worker * slen
            p_hi = min(min(P, (seg + 1) * 128), p_lo + slen)
            if p_hi <= p_lo:
                continue
            slots = [int(doc[t]) if 0 <= t < T else -1 for t in range(p_lo - 2, p_hi)]
            walked += 1 if len(set(slots)) == 1 else p_hi - p_lo
    return walked


@pytest.mark.parametrize('T', [100, 126, 127, 250, 1000])
def test_walked_positions_follow_the_kernels_slice_rule(T):
    rng = np.random.default_rng(T)
    docs = rng.integers(1, 50, size=(6, T))
    docs[0, :] = 0                                           # an all-padding document
    docs[1, T // 3:] = 0                                     # a padded tail
    docs[2, :] = 7                                           # one word repeated: uniform inside, not at the edges
    docs[3, 40:200] = 0                                      # padding in the middle
    w, total = bench.walked_positions(docs)
    assert total == 6 * (T + 2)
    assert w == sum(_walked_by_the_kernels_rule(d) for d in docs)
    w3, t3 = bench.walked_positions(docs.reshape(2, 3, T))   # NARRE's [B, R, W] documents
    assert (w3, t3) == (w, total)


def test_config_legs_ride_on_the_default_line_only():
    assert bench.config_legs_wanted(_args([]), dp_job=False)
    assert bench.config_legs_wanted(_args(['--gpus', '1', '--steps', '20', '--warmup', '5']), dp_job=False)
    assert not bench.config_legs_wanted(_args([]), dp_job=True)
    assert not bench.config_legs_wanted(_args(['--no-cpu-baseline']), dp_job=False)
    assert bench.config_legs_wanted(_args(['--no-cpu-baseline', '--config-legs']), dp_job=False)
    assert not bench.config_legs_wanted(_args(['--workload', 'cfg2_mfdot_electronics']), dp_job=False)
    assert not bench.config_legs_wanted(_args(['--doc-fill', 'full']), dp_job=False)
    labels = [l for l, _ in bench.CONFIG_LEGS]
    for want in ('cfg1_bias_only_musical', 'cfg2_mfdot_electronics', 'cfg2_mfdot_electronics_b8192', 'cfg4_narre_kindle',
                 'cfg5_transnetpp_synthetic', 'cfg3_full_uniform', 'cfg5_full_uniform_hbm_gather'):
        assert want in labels
