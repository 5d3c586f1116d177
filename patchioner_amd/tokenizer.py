"""Host-side detokeniser: CLIP BPE ids -> text (reference: P/src/clip/simple_tokenizer.py:129-132,
caller P/src/decap/decap.py:162-181).

Only *decoding* is on the captioning path.  The id -> byte-string table is a data asset derived from
the CLIP BPE vocabulary (``assets/clip_bpe_decode_table.npz``, produced by
``tools/oracle/gen_golden.py``): entry ``i`` is the UTF-8 byte string of sub-word ``i`` with the
end-of-word marker kept as the literal bytes ``</w>``; 49408 entries, the last two being
``<|startoftext|>`` and ``<|endoftext|>``.
"""
from __future__ import annotations

import os
from typing import List, Optional, Sequence

import numpy as np

_ASSET = os.path.join(os.path.dirname(os.path.abspath(__file__)), "assets", "clip_bpe_decode_table.npz")

EOT = "<|endoftext|>"
SOT = "<|startoftext|>"


class ClipDetokenizer:
    def __init__(self, table_path: Optional[str] = None):
        z = np.load(table_path or os.environ.get("PIO_BPE_TABLE", _ASSET))
        offs, blob = z["offsets"], z["blob"].tobytes()
        self.table = [blob[offs[i]:offs[i + 1]] for i in range(len(offs) - 1)]
        self.vocab_size = len(self.table)

    def decode(self, tokens: Sequence[int]) -> str:
        """Raises (IndexError) for ids outside the vocabulary, like the reference's KeyError."""
        tab = self.table
        for t in tokens:
            if t < 0 or t >= self.vocab_size:
                raise IndexError("token id %d outside the %d-entry CLIP vocabulary" % (t, self.vocab_size))
        raw = b"".join(tab[t] for t in tokens)
        return raw.decode("utf-8", errors="replace").replace("</w>", " ")

    def batch_captions(self, ids: Sequence[Sequence[int]], return_start_end_tokens: bool = False,
                       decoding_method=None) -> Optional[List[str]]:
        """Post-processing of ``decoding_batched`` (decap.py:162-181): cut at the first end-of-text,
        drop start-of-text; ANY undecodable row makes the whole batch ``None``."""
        try:
            outs = []
            for row in ids:
                row = [int(t) for t in row]
                s = decoding_method(row) if decoding_method is not None else self.decode(row)
                s = s.split(EOT)[0]
                if not return_start_end_tokens:
                    s = s.replace(SOT, "")
                else:
                    s += EOT
                outs.append(s)
            return outs
        except Exception:
            return None
