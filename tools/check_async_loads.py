"""ISA check for the hand-counted asynchronous loads of the GEMM epilogue (csrc/gemm256.hip).

The fast epilogue issues its side-operand / bias loads from inline asm and waits for them with hand-counted
`s_waitcnt vmcnt(N)`.  The compiler believes an asm output is valid the moment the asm statement ends: under register
pressure it may spill such a register (storing data that has not arrived) and hand the physical register to
something else -- the load then lands on top of the new owner (a pointer, say: a wild address and a GPU fault; seen
once, when unrelated code raised the kernel's pressure).  Nothing in the source can forbid that, so this script reads
the generated assembly: between an asm load and the wait that retires it, no instruction may write its destination
registers and no scratch store may read them.  vmcnt retires in issue order, so a load is retired by `vmcnt(N)` once at least N vector-memory operations
were issued after it.

usage: check_async_loads.py file.s      (hipcc --cuda-device-only -S ...)     exit status 1 on a violation"""
import re
import sys

VM = re.compile(r'^(global_load|global_store|global_atomic|buffer_load|buffer_store|buffer_atomic|scratch_load|scratch_store|flat_load|flat_store|flat_atomic)')
REG = re.compile(r'\bv(\d+)\b|\bv\[(\d+):(\d+)\]')


def regs_of(text):
    out = set()
    for m in REG.finditer(text):
        if m.group(1) is not None:
            out.add(int(m.group(1)))
        else:
            out.update(range(int(m.group(2)), int(m.group(3)) + 1))
    return out


def parse(lines):
    """-> list of (line no, kind, payload): kinds label / branch (target, conditional) / wait n / asmload regs /
    vm / other, with the registers an instruction touches."""
    out = []
    in_asm = False
    for no, raw in lines:
        t = raw.strip()
        if t.startswith(';;#ASMSTART'):
            in_asm = True
            continue
        if t.startswith(';;#ASMEND'):
            in_asm = False
            continue
        m = re.match(r'^(\.LBB\w+):', t)
        if m:
            out.append((no, 'label', m.group(1), t))
            continue
        if not t or t.startswith((';', '.')) or t.endswith(':'):
            continue
        ins = t.split(';')[0].strip()
        if not ins:
            continue
        m = re.match(r'^s_(c?branch\w*)\s+(\.LBB\w+)', ins)
        if m:
            out.append((no, 'branch', (m.group(2), m.group(1).startswith('cbranch')), ins))
            continue
        m = re.search(r's_waitcnt.*vmcnt\((\d+)\)', ins)
        if m:
            out.append((no, 'wait', int(m.group(1)), ins))
            continue
        if in_asm and ins.startswith('global_load'):
            out.append((no, 'asmload', frozenset(regs_of(ins.split(',')[0])), ins))
            continue
        # what matters: the registers an instruction WRITES (its first operand, unless it is a store) and what a
        # scratch store reads.  A plain read of a pending register shows up only on paths that cannot run (the wait is
        # under the same condition as the load: `if (bias) load ... if (bias) wait`), which this analysis does not know.
        if ins.startswith(('scratch_store', 'v_accvgpr_write')):
            touched = regs_of(ins)        # a spill (to scratch or to an AGPR) of data that may not have arrived
        else:
            touched = set()
        out.append((no, 'vm' if VM.match(ins) else 'other', frozenset(touched), ins))
    return out


def check_kernel(name, lines):
    """Forward dataflow over the kernel's control-flow graph.  State: {asm load (by line) -> fewest vector-memory
    operations issued after it on any path here}; a load leaves the state at a vmcnt(N) with N <= that count."""
    prog = parse(lines)
    label_at = {payload: i for i, (no, kind, payload, _) in enumerate(prog) if kind == 'label'}
    dest = {no: payload for no, kind, payload, _ in prog if kind == 'asmload'}
    text = {no: ins for no, kind, payload, ins in prog if kind == 'asmload'}
    state_in = {0: {}}
    work = [0]
    bad = {}
    CAP = 64

    def merge(i, st):
        old = state_in.get(i)
        if old is None:
            state_in[i] = dict(st)
            work.append(i)
            return
        changed = False
        for k, v in st.items():
            if k not in old or v < old[k]:
                old[k] = v
                changed = True
        if changed:
            work.append(i)

    while work:
        i = work.pop()
        st = dict(state_in[i])
        while i < len(prog):
            no, kind, payload, ins = prog[i]
            if kind == 'label':
                if i in state_in and state_in[i] is not st:
                    merge(i, st)
                    st = dict(state_in[i])
                else:
                    state_in[i] = dict(st)
            elif kind == 'branch':
                tgt, cond = payload
                if tgt in label_at:
                    merge(label_at[tgt], st)
                if not cond:
                    break
            elif kind == 'wait':
                st = {k: v for k, v in st.items() if v < payload}
            else:
                if kind != 'asmload':
                    for k in st:
                        if payload & dest[k]:
                            bad[(k, no)] = (name, k, text[k], no, ins)
                if kind in ('vm', 'asmload'):
                    st = {k: min(CAP, v + 1) for k, v in st.items()}
                if kind == 'asmload':
                    st[no] = 0
                if ins.startswith('s_endpgm'):
                    break
            i += 1
            if i < len(prog) and prog[i][1] == 'label':
                merge(i, st)
                break
    return list(bad.values())


def main(path):
    text = open(path).read().split('\n')
    kernels, cur, name = [], None, None
    for i, l in enumerate(text, 1):
        m = re.match(r'^(_ZN4vitk\w+):', l)
        if m:
            name, cur = m.group(1), []
            kernels.append((name, cur))
        elif cur is not None:
            cur.append((i, l))
            if 's_endpgm' in l:
                cur = None
    bad = []
    nloads = 0
    for name, lines in kernels:
        nloads += sum(1 for _, l in lines if l.strip().startswith('global_load_dwordx4') and ', off' in l)
        bad += check_kernel(name, lines)
    for name, lno, ltxt, no, ins in bad[:40]:
        print(f'{name}: register of the asynchronous load at line {lno} ({ltxt}) touched before its wait, line {no}: {ins}')
    print(f'{path}: {len(kernels)} kernels, {len(bad)} violation(s)')
    return 1 if bad else 0


if __name__ == '__main__':
    sys.exit(main(sys.argv[1]))
