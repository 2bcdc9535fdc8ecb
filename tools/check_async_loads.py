"""ISA check for the hand-counted asynchronous loads of the GEMM epilogue (csrc/gemm256.hip).

The fast epilogue issues its side-operand / bias loads from inline asm and waits for them with hand-counted
`s_waitcnt vmcnt(N)`.  The compiler believes an asm output is valid the moment the asm statement ends: under register
pressure it may spill such a register (storing data that has not arrived) and hand the physical register to
something else -- the load then lands on top of the new owner (a pointer, say: a wild address and a GPU fault; seen
once, when unrelated code raised the kernel's pressure).  Nothing in the source can forbid that, so this script reads
the generated assembly: between an asm load and the wait that retires it
  * no instruction may WRITE one of its destination registers (the first operand of every VALU, LDS-read and
    vector-memory load instruction counts as written; the register would be clobbered when the load lands), and
  * no instruction that only MOVES data may READ one: a spill to scratch (`scratch_store`), to an AGPR
    (`v_accvgpr_write`), a VGPR-to-VGPR copy (`v_mov_b32`, `v_mov_b64`, `v_pk_mov_b32`, `v_accvgpr_mov`), or a store to
    memory -- each of them saves bytes that have not arrived yet.
Arithmetic reads are not flagged: they show up only on paths that cannot run (the wait sits under the same condition
as the load, `if (bias) load ... if (bias) wait`, which a path-insensitive walk does not know); a copy on such a path
would be flagged all the same, so a report may need a look at the source -- the current kernels produce none.
vmcnt retires in issue order, so a load is retired by `vmcnt(N)` once at least N vector-memory operations were issued
after it.  A kernel body runs from its label to `.Lfunc_end` (not to the first `s_endpgm`: a kernel may have several),
and a file in which no asm load is found at all is an error (the check would pass vacuously).

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
        out.append((no, 'vm' if VM.match(ins) else 'other', touched_regs(ins), ins))
    return out


MOVES = ('v_mov_b32', 'v_mov_b64', 'v_pk_mov_b32', 'v_accvgpr_write', 'v_accvgpr_mov', 'v_swap_b32')
STORES = ('scratch_store', 'global_store', 'buffer_store', 'flat_store', 'ds_write', 'ds_store')
NO_VDST = ('v_cmp', 'v_cmpx', 'v_readlane', 'v_readfirstlane', 'v_nop', 's_', 'global_atomic', 'buffer_atomic',
           'flat_atomic', 'ds_add', 'ds_max', 'ds_min')        # (no-return atomics; a returning one carries a vdst first)


def _src_regs(args):
    parts = args.split(',')
    return regs_of(','.join(parts[1:])) if len(parts) > 1 else set()


def touched_regs(ins):
    """(spilled, copied, written): VGPRs an instruction must not touch while they are the destination of a pending
    asynchronous load.
      spilled: read by a spill (scratch store / AGPR write)      -> violation on every path, loop-carried ones included
      copied : read by a copy or by a store to memory            -> violation inside the iteration that issued the load: the
               compiler holds the value live here and moves bytes that have not arrived
      written: overwritten (first operand of a VALU / load op)   -> reported as a note, not a failure: the compiler holds the
               value DEAD on this path, i.e. the path by-passes the value's uses -- and every wait sits with its uses
               (`if (bias) load ... if (bias) { wait; use }`), so a path-insensitive walk meets these on paths that cannot
               run (the structurizer's flag-register paths from the fast epilogue into the split-K one, for instance)."""
    op = ins.split()[0]
    args = ins[len(op):]
    first = args.split(',')[0]
    none = frozenset()
    if op.startswith(('scratch_store', 'v_accvgpr_write')):
        return frozenset(regs_of(args)), none, none
    if op.startswith(STORES):
        return none, frozenset(regs_of(args)), none      # every operand of a store is read
    if op.startswith(MOVES):
        return none, frozenset(_src_regs(args)), frozenset(regs_of(first))
    if op.startswith(NO_VDST):
        if ('atomic' in op or op.startswith('ds_')) and (' glc' in ins or 'rtn' in op):
            return none, none, frozenset(regs_of(first))
        return none, none, none
    if op.startswith(('v_', 'global_load', 'buffer_load', 'scratch_load', 'flat_load', 'ds_read', 'ds_load', 'ds_bpermute',
                      'ds_permute', 'ds_swizzle')):
        return none, none, frozenset(regs_of(first))      # the destination (a wide one covers several registers)
    return none, none, none


def check_kernel(name, lines):
    """Forward dataflow over the kernel's control-flow graph.  State: {asm load (by line) -> fewest vector-memory
    operations issued after it on any path here}; a load leaves the state at a vmcnt(N) with N <= that count.
    The walk is path-insensitive: `if (c) load ... if (c) wait` has a path load -> no wait that cannot run, and on it the
    load stays pending around the item loop, where the NEXT iteration's own `else` arm (registers zeroed instead of loaded)
    legitimately rewrites the registers.  So a pending load that has travelled a back edge (a branch to an earlier label)
    keeps only the spill rule; writes, copies and stores of its registers are checked up to the end of the iteration
    that issued it -- where every feasible path has passed its wait (the K loop of the next item retires everything
    older through its own counted waits)."""
    prog = parse(lines)
    label_at = {payload: i for i, (no, kind, payload, _) in enumerate(prog) if kind == 'label'}
    dest = {no: payload for no, kind, payload, _ in prog if kind == 'asmload'}
    text = {no: ins for no, kind, payload, ins in prog if kind == 'asmload'}
    state_in = {0: {}}
    work = [0]
    bad = {}
    notes = {}
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
                    if label_at[tgt] <= i:        # back edge: the loads go on as loop-carried (negative key marks it)
                        merge(label_at[tgt], {-abs(k): v for k, v in st.items()})
                    else:
                        merge(label_at[tgt], st)
                if not cond:
                    break
            elif kind == 'wait':
                st = {k: v for k, v in st.items() if v < payload}
            else:
                if kind != 'asmload':
                    spilled, copied, written = payload
                    for k in st:
                        hit = (spilled | copied) if k > 0 else spilled
                        if hit & dest[abs(k)]:
                            bad[(abs(k), no)] = (name, abs(k), text[abs(k)], no, ins)
                        elif k > 0 and written & dest[k]:
                            notes[(k, no)] = (name, k, text[k], no, ins)
                if kind in ('vm', 'asmload'):
                    st = {k: min(CAP, v + 1) for k, v in st.items()}
                if kind == 'asmload':
                    st[no] = 0
                if ins.startswith(('s_endpgm', 's_setpc', 's_trap')):
                    break
            i += 1
            if i < len(prog) and prog[i][1] == 'label':
                merge(i, st)
                break
    return list(bad.values()), list(notes.values())


def main(path):
    text = open(path).read().split('\n')
    kernels, cur, name = [], None, None
    for i, l in enumerate(text, 1):
        m = re.match(r'^(_ZN4vitk\w+):', l)
        if m:
            name, cur = m.group(1), []
            kernels.append((name, cur))
        elif cur is not None:
            if re.match(r'^\.Lfunc_end\d+:', l):
                cur = None
            else:
                cur.append((i, l))
    bad, notes = [], []
    nloads = 0
    for name, lines in kernels:
        nloads += sum(1 for _, kind, _, _ in parse(lines) if kind == 'asmload')
        b, n = check_kernel(name, lines)
        bad += b
        notes += n
    for name, lno, ltxt, no, ins in bad[:40]:
        print(f'{name}: register of the asynchronous load at line {lno} ({ltxt}) spilled / copied / stored before its wait, line {no}: {ins}')
    for name, lno, ltxt, no, ins in notes[:40]:
        print(f'note: {name}: register of the asynchronous load at line {lno} ({ltxt}) is dead and reused at line {no} ({ins}) '
              'on a walk that by-passes its wait')
    print(f'{path}: {len(kernels)} kernels, {nloads} asynchronous asm loads, {len(bad)} violation(s), {len(notes)} note(s)')
    if nloads == 0 or not kernels:
        print(f'{path}: no kernel / no asm load found -- wrong file, or the parser no longer matches the assembly')
        return 2
    return 1 if bad else 0


if __name__ == '__main__':
    sys.exit(main(sys.argv[1]))
