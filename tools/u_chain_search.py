"""Which addition-subtraction chain for x^u (u = 0x44e992b44a6909f1, x cyclotomic: inverses are free)?

Searches chains of the shape "build a small dictionary of odd powers, then signed digits over the dictionary": every dictionary that
MAXM products and MAXS squarings can build from x (values up to VMAX) is enumerated, and for each the cheapest digit string is found
by dynamic programming (tools/gen_constants.py: dictionary_digits is the same recursion).  Cost model: a cyclotomic squaring 1, a
product 2.34 (3 453 against 8 070 instructions per lane on the lane pair).

    python tools/u_chain_search.py [MAXM MAXS]        default 2 7: a few seconds;  3 6 takes a minute

Result used by the kernels (3 6): dictionary {x^3, x^15, x^75}: 61 squarings + 13 products (width-4 windows: 63 + 16; gnark's unsigned
chain 62 + 17).  {x^17, x^35} reaches 62 + 13 with a two-entry table — which the compiler keeps in spilled registers (see
csrc/pairing29.hip.hpp: f12_expt_to).  Nothing found below 13 products up to (3 8).
"""
import sys
U = 0x44E992B44A6909F1
SQ, MUL = 1.0, 2.34
VMAX = 130
sys.setrecursionlimit(100000)


def best_digits(dictionary):
    memo = {}

    def best(v):
        if v in memo:
            return memo[v]
        if abs(v) in dictionary:
            memo[v] = (0.0, (v,))
            return memo[v]
        res = (1e9, ())
        if v % 2 == 0:
            c, ds = best(v // 2)
            res = (c + SQ, (0,) + ds)
        else:
            for d in dictionary:
                for s in (d, -d):
                    m = v - s
                    if m == 0 or m % 2 or abs(m // 2) >= abs(v):
                        continue
                    c, ds = best(m // 2)
                    if c + SQ + MUL < res[0]:
                        res = (c + SQ + MUL, (s,) + ds)
        memo[v] = res
        return res
    return best(U)


def dictionaries(max_mul, max_sq):
    seen = {}

    def rec(avail, cost, ops, nm, ns):
        key = frozenset(avail)
        if key in seen and seen[key][0] <= cost:
            return
        seen[key] = (cost, ops)
        av = sorted(avail)
        if ns < max_sq:
            for a in av:
                if 2 * a <= VMAX and 2 * a not in avail:
                    rec(avail | {2 * a}, cost + SQ, ops + (("sqr", a, 2 * a),), nm, ns + 1)
        if nm < max_mul:
            for i, a in enumerate(av):
                for b in av[i + 1:]:
                    for v, op in ((a + b, "mul"), (b - a, "div")):
                        if 0 < v <= VMAX and v % 2 and v not in avail:
                            rec(avail | {v}, cost + MUL, ops + ((op, a, b, v),), nm + 1, ns)
    rec(frozenset({1}), 0.0, (), 0, 0)
    return seen


def main():
    max_mul, max_sq = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (2, 7)
    out = []
    for key, (cost, ops) in dictionaries(max_mul, max_sq).items():
        d = tuple(v for v in sorted(key) if v % 2)
        c, ds = best_digits(d)
        out.append((c + cost, d, ds, ops))
    out.sort(key=lambda t: t[0])
    print("width-4 windows: 63 squarings + 16 products = %.2f" % (63 * SQ + 16 * MUL))
    for total, d, ds, ops in out[:5]:
        table_mul = sum(1 for op in ops if op[0] != "sqr")
        nz = sum(1 for x in ds if x) - 1
        used = sorted(set(abs(x) for x in ds if x))
        print("%.2f: %d squarings + %d products; digits over %s" % (total, len(ds) - 1 + len(ops) - table_mul, nz + table_mul, used))
        print("   table:", ops)
        print("   digits, LSB first:", ds)


if __name__ == "__main__":
    main()
