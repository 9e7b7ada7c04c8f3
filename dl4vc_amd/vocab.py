"""Token vocabulary of the pileup encoding (reference: dl4vc/base_enum.py:7-27).

Ten tokens: 0 pad, 1 A, 2 T, 3 G, 4 C, 5 gap/N ('-'), 6 read start, 7 read end,
8 'noinsert' (read has nothing where another allele inserts), 9 unknown/IUPAC.
"""
PAD, A, T, G, C, GAP, START, END, NOINSERT, UNK = range(10)
VOCAB_SIZE = 10
TOKEN_NAMES = {PAD: "p", A: "A", T: "T", G: "G", C: "C", GAP: "-", START: "s", END: "e",
               NOINSERT: "noinsert", UNK: "?"}

# character -> token (base_enum.py:7-11).  's' maps to START first and is then overwritten by the
# IUPAC 'S'/'s' -> 9 entry in the reference's dict literal (later key wins), which we keep.
_CHAR_TOKEN = {}
for _chars, _tok in (("Aa", A), ("TtUu", T), ("Gg", G), ("Cc", C), ("-*NnXx.", GAP),
                     ("e", END), ("?MmKkRrYySsWwBbVvHhDd", UNK)):
    for _c in _chars:
        _CHAR_TOKEN[_c] = _tok
_CHAR_TOKEN[""] = GAP

# strand enum of a pileup column (base_enum.py:15-23): lower-case base = reverse strand
STRAND_PAD, STRAND_LOWER, STRAND_UPPER = 0, 1, 2

# base_enum.py:25 -- the set the reference uses to recognise a SNP; note it is CHARACTERS and it
# lacks lower-case 'g' (a quirk we keep: 'g'->X is not classified as a SNP there either).
SNP_BASE_CHARS = frozenset("AaTtCcG")

# dl4vc/base_enum.py:27
MUTATION_SNP, MUTATION_INSERT, MUTATION_DELETE, MUTATION_UNKNOWN = 1, 2, 3, 0


def token_of(ch: str) -> int:
    """Token for one allele character; KeyError for characters the reference's table lacks."""
    return _CHAR_TOKEN[ch]
