"""Prints DESIGN.md section 5's table of records from a bench full record (profiles/rNN_bench.json)."""
import json
import sys

d = json.load(open(sys.argv[1] if len(sys.argv) > 1 else 'profiles/r06_bench.json'))
sub = d['sub']


def row(name, r, ref=None):
    rf = r.get('roofline') or {}
    cells = [name, '%.1f' % r['value'], '%.3f' % r['ms_per_step'] + (' = %.3f ×' % r['ms_per_step_vs_plain'] if r.get('ms_per_step_vs_plain') else ''),
             '%.1f' % rf['launches_per_step'] if rf.get('launches_per_step') else '', '%.1f (%.3f)' % (rf['achieved'], rf['frac']) if rf.get('achieved') else '',
             '%.1f ms' % r['host_issue_ms_per_step_empty_queue'] if r.get('host_issue_ms_per_step_empty_queue') else '',
             '%.0f×' % r['gpu_vs_cpu'] if r.get('gpu_vs_cpu') else '']
    print('| ' + ' | '.join(cells) + ' |')


print('| record | steps/s | ms/step | GEMM launches/step | GEMM TF/s (frac of 157.3) | host issue, empty queue | GPU/CPU port |')
print('|---|---|---|---|---|---|---|')
row('supernet weight step (configs[2], headline)', d)
for k, n in (('arch_step', "arch step, MODE 'full'"), ('bilevel', 'bilevel round (5+1, optimizers in)'), ('train_vqa', 'train_vqa (configs[1])'),
             ('search_vqa_stream', 'search_vqa_stream (fresh host batch per step)'), ('search_vqa_dropin', 'search_vqa_dropin (`search_vqa.py:279-301` unchanged, torch Adam)'),
             ('search_vqa_dp1', 'search_vqa_dp1 (exchange in a one-rank RCCL group)'), ('train_vqa_dp1', 'train_vqa_dp1'),
             ('search_vqa_unpad', '**search_vqa_unpad** (ragged stream; flops on the valid rows)'), ('train_vqa_unpad', '**train_vqa_unpad**')):
    row(n, sub[k])
print()
print('kernel classes (ms/step):', {k: round(v['ms_per_step'], 3) for k, v in d['kernel_classes'].items()})
print('blocks:', d.get('blocks_ms_per_step'))
