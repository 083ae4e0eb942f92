"""Instruction counts of one kernel of a hipcc -save-temps assembly file: python scripts/isa_count.py file.s <name substring> ..."""
import sys
s = open(sys.argv[1]).read().split('\n')
for k in sys.argv[2:]:
    start = next((i for i, l in enumerate(s) if l.startswith('_Z') and k in l.split(':')[0] and ':' in l), None)
    if start is None:
        print(k, 'not found'); continue
    end = next(j for j in range(start, len(s)) if 's_endpgm' in s[j])
    body = s[start:end]
    c = lambda p: sum(1 for l in body if l.strip().startswith(p))
    print(k, 'lines', len(body), 'mfma', c('v_mfma'), 'v_exp', c('v_exp_f32'), 'ds_read', c('ds_read'), 'ds_write', c('ds_write'),
          'barrier', c('s_barrier'), 'global_load', c('global_load'), 'global_store', c('global_store'), 'scratch', c('scratch_'))
    open('/tmp/k_%s.s' % k.replace('<', '_').replace('>', '_'), 'w').write('\n'.join(body))
