"""order of the rollout's LDS stores in the compiler's assembly of an mfmat kernel (hipcc -S of one instantiation):
the mask-free stores of the hand-over (admm_mfmat.hip.h) land on cells another lane owns, and the owner's store has to come
later in program order — an order between lanes the compiler cannot see.  Decodes every ds_write of the rollout (base
register + immediates), checks (a) u_{k-1}'s store (whose spare lanes hit x_k's first rows) before x_k's slot-0 store and
(b) x_k's slot-1 store (spare lanes: u_k's cells) before u_k's store.    usage: check_handover_order.py file.s NX NU N"""
import re, sys


def kernels(path):
    """the assembly of every kernel of the file that has matrix-core products"""
    out, cur = [], []
    for l in open(path).read().split("\n"):
        cur.append(l)
        if "s_endpgm" in l:
            if any("v_mfma" in c for c in cur):
                out.append(cur)
            cur = []
    return out


def decode(lines, NX, NU, N):
    idx = [i for i, l in enumerate(lines) if "v_mfma" in l]
    start, end = max(idx[0] - 60, 0), None
    for a, b in zip(idx, idx[1:]):
        if b - a > 150:
            end = a + 80
            break
    first = None
    for i in range(start, end):
        m = re.match(r"(ds_write\w*) (v\d+), (.*)", lines[i].strip())
        if m:
            first = m.group(2)
            break
    base, writes = {first: 0}, []
    for i in range(start, end):
        l = lines[i].strip()
        m = re.match(r"v_add_u32_e32 (v\d+), (0x[0-9a-f]+|\d+), (v\d+)", l)
        if m:
            d, c, s = m.groups()
            if s in base:
                base[d] = base[s] + int(c, 0)
            elif d in base and d != first:
                del base[d]
            continue
        m = re.match(r"(ds_write\w*) (v\d+), (.*)", l)
        if m:
            op, addr, rest = m.groups()
            if addr not in base:
                continue
            o = base[addr]
            g = lambda n: int(re.search(n + r":(0x[0-9a-fA-F]+|\d+)", rest).group(1), 0) if re.search(n + r":(0x[0-9a-fA-F]+|\d+)", rest) else 0   # (inline asm prints hex)
            if op == "ds_write_b32":
                writes.append((i, o + g("offset")))
            elif op == "ds_write2_b32":
                writes += [(i, o + 4 * g("offset0")), (i + 0.5, o + 4 * g("offset1"))]
            elif op == "ds_write2st64_b32":
                writes += [(i, o + 256 * g("offset0")), (i + 0.5, o + 256 * g("offset1"))]
            continue
        m = re.match(r"\w+ (v\d+),", l)
        if m and m.group(1) in base and m.group(1) != first:
            del base[m.group(1)]
    plen, u0 = 64 * (NX + NU), 64 * NX
    pos = {}
    for ln, a in writes:
        pos.setdefault(a, ln)
    expected = {k * plen for k in range(N)} | {k * plen + 256 for k in range(N)} | {k * plen + u0 - plen for k in range(1, N)}
    bad = []
    for k in range(1, N):
        if pos.get(k * plen + u0 - plen, 0) > pos.get(k * plen, 1e9):
            bad.append(("u_{k-1} after x_k", k))
        if k < N - 1 and pos.get(k * plen + 256, 0) > pos.get((k + 1) * plen + u0 - plen, 1e9):
            bad.append(("x_k slot 1 after u_k", k))
    return sorted(expected - set(pos)), bad


if __name__ == "__main__":
    rc = 0
    for lines in kernels(sys.argv[1]):
        missing, bad = decode(lines, *(int(a) for a in sys.argv[2:5]))
        print("stores not found:", missing, " out of order:", bad)
        rc |= 1 if missing or bad else 0
    sys.exit(rc)
