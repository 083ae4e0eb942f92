"""Audit of kernels whose global loads sit in inline asm behind hand-counted waits (cdna_hip_programming.md 5.7): hipcc does not know
that an asm load's destination is written LATER, so between the load and the counted `s_waitcnt vmcnt(N)` that covers it no
compiler-generated instruction may read or write that register.  The kernel's instructions are walked in program order; every
backward branch re-walks its loop body once more with the pending set carried around the back edge (so a load of iteration c that
an instruction of iteration c + 1 touches is seen).  `python scripts/isa_asm_load_audit.py file.s <kernel name substring> ...`
(file.s from `hipcc -save-temps`); exit status 1 on a violation.  Used by tests/test_kernel_isa_cpu.py for the f32 GEMMs."""
import re
import sys


def _regs(tok):
    m = re.match(r"v\[(\d+):(\d+)\]", tok)
    if m:
        return set(range(int(m.group(1)), int(m.group(2)) + 1))
    m = re.match(r"v(\d+)$", tok)
    return {int(m.group(1))} if m else set()


def kernel_body(lines, key):
    start = next((i for i, l in enumerate(lines) if l.startswith('_Z') and key in l.split(':')[0] and ':' in l), None)
    if start is None:
        return None
    end = next(j for j in range(start, len(lines)) if 's_endpgm' in lines[j])
    return lines[start:end + 1]


def audit(body):
    """returns (violations, n_asm_loads, n_counted_waits)"""
    labels = {l.split(':')[0].strip(): i for i, l in enumerate(body) if re.match(r"^\.LBB\S+:", l)}
    viol, state = [], dict(pending={}, issued=0)
    n_loads = n_waits = 0

    def walk(lo, hi, depth):
        nonlocal n_loads, n_waits
        in_asm = False
        i = lo
        while i < hi:
            t = body[i].strip()
            i += 1
            if t.startswith(';;#ASMSTART'):
                in_asm = True
                continue
            if t.startswith(';;#ASMEND'):
                in_asm = False
                continue
            if not t or t[0] in ';.' or t.endswith(':'):
                continue
            if in_asm and t.startswith('global_load_dword'):
                for r in _regs(t.split()[1].rstrip(',')):
                    state['pending'][r] = state['issued']
                state['issued'] += 1
                if depth == 0:
                    n_loads += 1
                continue
            if in_asm and t.startswith('s_waitcnt') and 'vmcnt' in t:
                n = int(re.search(r'vmcnt\((\d+)\)', t).group(1))
                done = state['issued'] - n
                for r in [r for r, k in state['pending'].items() if k < done]:
                    del state['pending'][r]
                if depth == 0:
                    n_waits += 1
                continue
            if in_asm:
                continue
            if t.startswith('s_waitcnt') and 'vmcnt(0)' in t:
                state['pending'].clear()
                continue
            used = set()
            for tk in re.findall(r"v\[\d+:\d+\]|\bv\d+\b", t):
                used |= _regs(tk)
            hit = used & set(state['pending'])
            if hit:
                viol.append((i - 1, t, sorted(hit)[:4]))
            m = re.match(r"s_cbranch_\w+\s+(\.LBB\S+)", t) or re.match(r"s_branch\s+(\.LBB\S+)", t)
            if m and depth < 2 and m.group(1) in labels and labels[m.group(1)] < i - 1:
                walk(labels[m.group(1)], i - 1, depth + 1)        # the loop body once more, pending set carried over the back edge
    walk(0, len(body), 0)
    return viol, n_loads, n_waits


if __name__ == "__main__":
    lines = open(sys.argv[1]).read().split('\n')
    bad = False
    for key in sys.argv[2:]:
        body = kernel_body(lines, key)
        if body is None:
            print(key, "not found")
            bad = True
            continue
        viol, nl, nw = audit(body)
        print(f"{key}: {nl} asm loads, {nw} counted waits, {len(viol)} violations")
        for v in viol[:8]:
            print("   ", v)
        bad |= bool(viol)
    sys.exit(1 if bad else 0)
