"""CPU model of the pathing index's table logic (common.h index_find / exact_find, step2_graph.hip k_index_insert / k_index_mark /
k_exact_insert): open addressing with linear probing; a stored key gives its bit 0 to the strand and keeps it in a spare bit of the slot; the
entries of a key with more than IDX_HARD entries are marked and every k-mer they serve lives in an exact table, where a lookup that meets a
marked entry goes and stays.  The model checks the property the kernels rely on -- a lookup finds its k-mer's entry if and only if it
exists, whatever shares its probe sequence -- on keys chosen to collide: equal keys (repeat copies), keys that differ in bit 0 only, keys
with one home slot."""
import numpy as np
import pytest

IDX_HARD = 3
CAP = 256                                     # slots (a power of two)


def home(key):                                # stands for bucket_mix(key) & mask: keys that differ in bit 0 have unrelated homes
    return ((key * 0x85EBCA6B) & 0xFFFFFFFF) >> 24 & (CAP - 1)


def home_twins_together(key):                 # the worst case for the redirect: both keys of a twin pair in ONE run of slots
    return home(key & ~1)


class Index:
    """slot = (key with bit 0 replaced by the strand, the key's own bit 0, unipath, position, marked)"""
    def __init__(self, keep_bit0, home=home):
        self.home = home
        self.slots = [None] * CAP
        self.keep_bit0 = keep_bit0            # False: the first form of the exact table's redirect (a marked entry of the twin key misleads)
        self.exact = {}                       # k-mer (here: its position) -> unipath

    def insert(self, key, strand, unipath, pos):
        s = self.home(key)
        while self.slots[s] is not None: s = (s + 1) & (CAP - 1)
        self.slots[s] = [(key & ~1) | strand, key & 1, unipath, pos, False]

    def same_key(self, v, key):
        return (v[0] ^ key) & ~1 == 0 and (not self.keep_bit0 or v[1] == (key & 1))

    def mark(self, entries, kmers_of):
        """k_index_hard_list + k_index_hard_apply: an entry whose key has more than IDX_HARD entries along its probe sequence is marked, the k-mers it
        serves go to the exact table"""
        for key, strand, unipath, pos in entries:
            n, s = 0, self.home(key)
            while self.slots[s] is not None:
                n += self.same_key(self.slots[s], key)
                s = (s + 1) & (CAP - 1)
            if n > IDX_HARD:
                s = self.home(key)
                while True:
                    v = self.slots[s]
                    if v[0] == ((key & ~1) | strand) and v[2] == unipath and v[3] == pos: v[4] = True; break
                    s = (s + 1) & (CAP - 1)
                for km in kmers_of(pos): self.exact[km] = unipath

    def find(self, key, kmer, verify):
        """index_find: the first entry of the key decides -- a marked one sends the lookup to the exact table, an unmarked one is verified"""
        s, probes = self.home(key), 0
        while self.slots[s] is not None:
            v = self.slots[s]; probes += 1
            if self.same_key(v, key):
                if v[4]: return self.exact.get(kmer), probes
                if verify(v, kmer): return v[2], probes
            s = (s + 1) & (CAP - 1)
        return None, probes


def make_case(rng, twins):
    """entries: groups of equal keys of sizes 1 .. 12 (copies of a repeat), every entry serving the k-mers pos*8 .. pos*8+3; twins: for some
    groups a second group whose key differs in bit 0 only, forced into the same run of slots"""
    entries, pos = [], 0
    for g in range(14):
        key = int(rng.integers(0, 1 << 32)) & ~1
        for _ in range(int(rng.integers(1, 13))):
            entries.append((key, int(rng.integers(0, 2)), len(entries), pos)); pos += 1
        if twins and g % 3 == 0:
            for _ in range(int(rng.integers(1, 4))):        # a small (unmarked) group under the twin key
                entries.append((key | 1, int(rng.integers(0, 2)), len(entries), pos)); pos += 1
    return entries


@pytest.mark.parametrize("seed", range(6))
@pytest.mark.parametrize("twins", [False, True])
def test_every_kmer_is_found_exactly_where_it_is(seed, twins):
    rng = np.random.default_rng(seed)
    entries = make_case(rng, twins)
    kmers_of = lambda pos: [pos * 8 + t for t in range(4)]
    X = Index(keep_bit0=True, home=home_twins_together if twins else home)
    order = rng.permutation(len(entries))
    for i in order: X.insert(*entries[i])
    X.mark(entries, kmers_of)
    verify = lambda v, kmer: kmer // 8 == v[3] and kmer % 8 < 4
    for key, strand, unipath, pos in entries:
        for km in kmers_of(pos):
            assert X.find(key, km, verify)[0] == unipath
        assert X.find(key, pos * 8 + 7, verify)[0] is None   # a k-mer that is in no unipath (a sequencing error beside this minimizer)


def test_without_the_keys_bit_0_a_twin_misleads():
    """the first form of the redirect: the stored key has lost bit 0.  Six entries of key K run from its home slot 5 to slot 10; the one entry of
    the twin key K | 1 (home slot 9) lies behind them.  Counted from ITS home the twin sees three entries of "its" key -- not marked --, but a lookup
    from slot 9 meets a marked entry of K first and asks the exact table, which does not have the twin's k-mers."""
    homes = {0x1000: 5, 0x1001: 9}
    entries = [(0x1000, 0, i, i) for i in range(6)] + [(0x1001, 0, 6, 6)]
    kmers_of = lambda pos: [pos * 8 + t for t in range(4)]
    verify = lambda v, kmer: kmer // 8 == v[3] and kmer % 8 < 4
    bad = Index(keep_bit0=False, home=homes.get)
    for e in entries: bad.insert(*e)
    bad.mark(entries, kmers_of)
    assert bad.find(0x1001, 6 * 8, verify)[0] is None        # missed
    good = Index(keep_bit0=True, home=homes.get)
    for e in entries: good.insert(*e)
    good.mark(entries, kmers_of)
    assert good.find(0x1001, 6 * 8, verify)[0] == 6
    assert all(good.find(0x1000, i * 8 + t, verify)[0] == i for i in range(6) for t in range(4))
