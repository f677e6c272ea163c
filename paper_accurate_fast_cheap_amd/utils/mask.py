"""Padding masks (reference: wenet/utils/mask.py:200-253).  Only what the RWKV encoder path needs:
the paper's configs use neither dynamic nor static chunk masks, for which add_optional_chunk_mask
(mask.py:126-197) returns its input unchanged."""
import torch


def make_pad_mask(lengths: torch.Tensor, max_len: int = 0) -> torch.Tensor:
    """(B,) lengths -> (B, max_len) bool, True on PADDED positions (mask.py:200-226)."""
    max_len = max_len if max_len > 0 else int(lengths.max().item())
    seq = torch.arange(0, max_len, dtype=torch.int64, device=lengths.device)
    return seq.unsqueeze(0) >= lengths.unsqueeze(-1)


def make_non_pad_mask(lengths: torch.Tensor) -> torch.Tensor:
    return ~make_pad_mask(lengths)


def add_optional_chunk_mask(xs, masks, use_dynamic_chunk: bool, use_dynamic_left_chunk: bool,
                            decoding_chunk_size: int, static_chunk_size: int, num_decoding_left_chunks: int):
    """Identity for the recurrent-attention slot: the slot never reads `mask` (rwkv_wrapper.py:57-83), and the
    paper's YAMLs leave use_dynamic_chunk / static_chunk_size off.  Chunk-limited attention masks belong to the
    MHA baseline, which is out of scope; asking for them is an error rather than a silent no-op."""
    if use_dynamic_chunk or static_chunk_size > 0:
        raise NotImplementedError("chunk attention masks apply to the MHA encoder, not to the recurrent slot")
    return masks
