"""Host-side behaviour of bench.py that needs no GPU."""
import importlib.util
import os
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench():
    spec = importlib.util.spec_from_file_location('bench_module', os.path.join(ROOT, 'bench.py'))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


def test_self_spawn_stops_the_other_ranks_when_one_fails(tmp_path, capfd):
    """A rank that exits non-zero must not leave `bench.py --gpus N` waiting on ranks parked in a collective."""
    script = tmp_path / 'rank.py'
    script.write_text("import os, sys, time\n"
                      "if os.environ['RANK'] == '1':\n"
                      "    sys.exit(3)\n"
                      "print('{\"rank0\": true}', flush=True)\n"
                      "time.sleep(600)\n")
    t0 = time.time()
    rc = _bench()._self_spawn(2, [], script=str(script))
    assert rc == 3
    assert time.time() - t0 < 60
    out = capfd.readouterr()
    assert 'rank0' in out.out and 'stopping the other ranks' in out.err


def test_self_spawn_relays_rank0_and_returns_zero(tmp_path, capfd):
    script = tmp_path / 'rank.py'
    script.write_text("import os\nif os.environ['RANK'] == '0':\n    print('{\"n\": %s}' % os.environ['WORLD_SIZE'])\n")
    assert _bench()._self_spawn(2, [], script=str(script)) == 0
    assert '{"n": 2}' in capfd.readouterr().out


def test_self_spawn_has_a_wall_clock_limit(tmp_path, monkeypatch, capfd):
    script = tmp_path / 'rank.py'
    script.write_text("import time\ntime.sleep(600)\n")
    monkeypatch.setenv('MMNAS_BENCH_SPAWN_TIMEOUT', '2')
    t0 = time.time()
    assert _bench()._self_spawn(2, [], script=str(script)) == 124
    assert time.time() - t0 < 60


def test_vgd_bce_loss_average_defaults_to_the_batch_rows():
    """train_vgd.py:316-333 divides the BCE term by the batch size; left out, it is the leading dimension."""
    import torch
    import torch.nn.functional as F
    from mmnas_amd.harness import vgd_loss
    g = torch.Generator().manual_seed(5)
    ps, pr = torch.randn(6, 10, generator=g), torch.randn(6, 10, 4, generator=g)
    sc, bb = torch.rand(6, 10, generator=g), torch.rand(6, 10, 4, generator=g)
    sm, bm = torch.ones(6, 10), torch.ones(6, 10, 4)
    got = vgd_loss(ps, pr, sc, sm, bb, bm, scores_loss='bce')
    want = F.binary_cross_entropy_with_logits(ps, sc, reduction='sum') / 6 + 0.5 * F.smooth_l1_loss(pr, bb, reduction='sum') / bm.sum()
    assert torch.allclose(got, want)
    assert torch.allclose(vgd_loss(ps, pr, sc, sm, bb, bm, scores_loss='bce', batch_size=3),
                          want + F.binary_cross_entropy_with_logits(ps, sc, reduction='sum') / 6)


def test_compact_line_of_a_full_default_record_stays_small():
    """The driver keeps ~16 KB of stdout and parses the last line: the line made from a full `--workload all` record
    (seven sub records, kernel classes, PMC detail) must stay under the limit and keep what the driver reads."""
    import json
    b = _bench()
    rec = json.load(open(os.path.join(ROOT, 'profiles', 'r03_bench.json')))
    assert len(json.dumps(rec)) > 16000          # the record that cost round 3 its parse
    line = json.dumps(b.compact_line(rec, 'gpurun_out/bench_full.json'), separators=(',', ':'))
    assert len(line) < 4500 < b.LINE_LIMIT
    d = json.loads(line)
    for k in ('metric', 'value', 'unit', 'n_gpus', 'steps', 'warmup', 'ms_per_step', 'higher_is_better', 'scaling', 'vs_baseline',
              'dtype', 'data', 'config', 'roofline', 'cpu_baseline', 'sub', 'full_record'):
        assert k in d, k
    assert set(d['roofline']) >= {'bound', 'achieved', 'peak', 'unit', 'frac', 'traffic'}
    assert set(d['cpu_baseline']) >= {'value', 'unit', 'cores', 'kind', 'sample'}
    assert set(d['sub']) == set(rec['sub']) and all('ms_per_step' in s for s in d['sub'].values())
    assert abs(d['value'] - rec['value']) < 1e-3


def test_fit_line_drops_optional_detail_instead_of_failing():
    """A record that grew past the limit must still give the driver its line (the old assert lost the run)."""
    import json
    b = _bench()
    rec = json.load(open(os.path.join(ROOT, 'profiles', 'r04_bench.json')))
    obj = b.compact_line(rec, 'gpurun_out/bench_full.json')
    obj['kernel_ms_per_step'] = {'class_%d' % i: 0.001 * i for i in range(600)}      # ~10 KB of optional detail
    line = b.fit_line(obj)
    assert len(line) < b.LINE_LIMIT
    d = json.loads(line)
    assert d['dropped_for_length'][0] == 'kernel_ms_per_step' and 'kernel_ms_per_step' not in d
    for k in ('metric', 'value', 'unit', 'n_gpus', 'ms_per_step', 'config', 'roofline', 'cpu_baseline', 'full_record'):
        assert k in d, k
    small = b.fit_line(b.compact_line(rec, 'x'))
    assert 'dropped_for_length' not in json.loads(small)


def test_summarize_prof_counts_its_steps_in_the_trace():
    """the r04 summaries divided by a literal 23 for a 26-step command.  The count now comes from the trace."""
    import importlib.util
    spec = importlib.util.spec_from_file_location('summarize_prof', os.path.join(ROOT, 'tools', 'summarize_prof.py'))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    import csv
    rows = list(csv.DictReader(open(os.path.join(ROOT, 'profiles', 'r04_search_vqa_kernel_stats.csv'))))
    steps, how = m.steps_from_trace(rows, 'auto')
    assert steps == 26 and 'lstm_seq_fwd_kernel' in how
    assert m.steps_from_trace(rows, '23')[0] == 26          # a literal that disagrees with the trace loses
    assert m.steps_from_trace(rows, 'auto/2')[0] == 13


def test_compact_line_keeps_what_limits_the_gemm_class():
    """the line says what LIMITS the d = 256 GEMM class -- `limited_by`, the measured `fixed_cost_share`
    and per-launch fixed cost -- with `frac_bf16_div6` next to `frac`, and who set the kernel-argument placement; checked on the
    round's committed full record (the GPU suite asserts the same on a live run)."""
    import json
    b = _bench()
    rec = json.load(open(os.path.join(ROOT, 'profiles', 'r06_bench.json')))
    line = json.dumps(b.compact_line(rec, 'gpurun_out/bench_full.json'), separators=(',', ':'))
    assert len(line) < b.LINE_LIMIT
    rl = json.loads(line)['roofline']
    keys = list(rl)
    assert rl['bound'] == 'mfma' and rl['limited_by'] == 'latency / fixed cost'
    assert 0.05 < rl['fixed_cost_share'] < 0.5 and 2.0 < rl['fixed_cost_us_per_launch'] < 8.0
    assert keys.index('frac_bf16_div6') == keys.index('frac') + 1 and 0 < rl['frac_bf16_div6'] < rl['frac'] < 1
    # share = launches per step x fixed cost / the class's time per step, all three in the record
    gm = rec['kernel_classes']['gemm']
    assert abs(rl['fixed_cost_share'] - rl['launches_per_step'] * rl['fixed_cost_us_per_launch'] * 1e-3 / gm['ms_per_step']) < 2e-3
    assert json.loads(line)['config']['hip_force_dev_kernarg_source'] in ('set_by_library', 'inherited')
