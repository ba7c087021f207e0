"""tests/cases.py — seeded toy inputs shared by the CPU and GPU test suites.
Own tiny PRNG (SplitMix64) so fixtures never depend on Python's `random`."""


class SplitMix:
    def __init__(self, seed):
        self.s = seed & 0xFFFFFFFFFFFFFFFF

    def next(self):
        self.s = (self.s + 0x9E3779B97F4A7C15) & 0xFFFFFFFFFFFFFFFF
        z = self.s
        z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & 0xFFFFFFFFFFFFFFFF
        z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & 0xFFFFFFFFFFFFFFFF
        return z ^ (z >> 31)

    def randint(self, lo, hi):
        return lo + self.next() % (hi - lo + 1)

    def choice(self, seq):
        return seq[self.next() % len(seq)]

    def random(self):
        return (self.next() >> 11) / float(1 << 53)


def random_dna(rng, n):
    return "".join("ACGT"[rng.next() & 3] for _ in range(n))


def toy_genome(seed, length, k, repeats=0, tandem=0, inverted=0, snp_every=0):
    """Returns list of read sequences (haplotypes) forming the DBG, first one is the
    genome gaps are cut from.  repeats: dispersed exact copies; tandem: tandem
    arrays (cycles -> non-trivial SCCs); inverted: reverse-complement copies
    (Q7 territory); snp_every: second haplotype with substitutions (bubbles)."""
    rng = SplitMix(seed * 7919 + 13)
    g = list(random_dna(rng, length))
    comp = {"A": "T", "C": "G", "G": "C", "T": "A"}
    for _ in range(repeats):
        ln = rng.randint(k + 2, 4 * k)
        src = rng.randint(0, length - ln - 1)
        dst = rng.randint(0, length - ln - 1)
        g[dst:dst + ln] = g[src:src + ln]
    for _ in range(tandem):
        unit = rng.randint(max(2, k // 2), 2 * k)
        copies = rng.randint(3, 6)
        src = rng.randint(0, length - unit * copies - 1)
        u = g[src:src + unit]
        for c in range(copies):
            g[src + c * unit:src + (c + 1) * unit] = u
    for _ in range(inverted):
        ln = rng.randint(k + 2, 3 * k)
        src = rng.randint(0, length - ln - 1)
        dst = rng.randint(0, length - ln - 1)
        g[dst:dst + ln] = [comp[c] for c in reversed(g[src:src + ln])]
    g = "".join(g)
    seqs = [g]
    if snp_every:
        h = list(g)
        p = snp_every // 2
        while p < length:
            h[p] = rng.choice([c for c in "ACGT" if c != h[p]])
            p += snp_every + rng.randint(0, 5)
        seqs.append("".join(h))
    return seqs


def one_strand_genome(seed, length, alphabet="AC"):
    """A genome over A and C only: the reverse complement of every k-mer is made of T and G, so no k-mer ever
    meets its own reverse strand and NO gap is a Q7 case — at k = 5 or 7 the toy genomes of
    toy_genome() make nearly every gap one, and Q7 gaps are outside the bit-exact comparison.  With 2^k
    possible k-mers the graph is dense: dozens of cycles per closure, path counts that saturate at MAX_PATHS."""
    rng = SplitMix(seed * 7919 + 13)
    return "".join(alphabet[rng.randint(0, len(alphabet) - 1)] for _ in range(length))


def cut_gaps(seed, genome, k, fuz, ngaps, min_len, max_len, d_err, vary_fuz=True, claim_noise=True):
    """List of dicts(left,right,gap_len,lmf,rmf,true_len).  The claimed gap length is
    chosen so that most gaps are fillable: the DP accepts path lengths
    g + lmf + j +- err with err <= d_err while the true path is true_len + k + lmf
    long (Gap2Seq.cpp:1125-1126), so claimed = true_len + k + noise."""
    rng = SplitMix(seed * 104729 + 7)
    out = []
    n = len(genome)
    for _ in range(ngaps):
        gl = rng.randint(min_len, max_len)
        lmf = rmf = fuz
        if vary_fuz and rng.random() < 0.25:
            rmf = rng.randint(0, fuz)
        if vary_fuz and rng.random() < 0.25:
            lmf = rng.randint(0, fuz)
        pos = rng.randint(k + lmf, n - gl - k - rmf - 1)
        claimed = gl
        if claim_noise:
            claimed = gl + k + rng.choice([0, 0, 0, -d_err, d_err, -(d_err // 2), d_err // 2, d_err + 1, -(d_err + 1), -k])
        claimed = max(1, claimed)
        out.append(dict(left=genome[pos - k - lmf:pos], right=genome[pos + gl:pos + gl + k + rmf], gap_len=claimed,
                        lmf=lmf, rmf=rmf, true_len=gl))
    return out


def scaffold_record(genome, k, fuz, pos_len_list, pad=5):
    """One multi-gap scaffold record: genome slice with N runs at the given
    (start, length, n_count) triples (sorted, non overlapping)."""
    lo = max(0, pos_len_list[0][0] - k - fuz - pad)
    hi = min(len(genome), pos_len_list[-1][0] + pos_len_list[-1][1] + k + fuz + pad)
    s = list(genome[lo:hi])
    out = []
    cur = lo
    for (p, ln, ncount) in pos_len_list:
        out.append(genome[cur:p])
        out.append("N" * ncount)
        cur = p + ln
    out.append(genome[cur:hi])
    return "".join(out)
