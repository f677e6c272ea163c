"""MI355X-native hot path of revdotcom/paper_accurate_fast_cheap: the bidirectional
recurrent-attention (RWKV-v6 WKV) Conformer encoder, behind the reference's plugin surface.
See DESIGN.md and INTEGRATION.md."""
__version__ = "0.1.0"
