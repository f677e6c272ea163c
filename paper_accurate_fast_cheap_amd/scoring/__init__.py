from .wer import ErrorCounts, WerScorer, align, characterize, giga_post_process, normalize, score_files

__all__ = ["ErrorCounts", "WerScorer", "align", "characterize", "giga_post_process", "normalize", "score_files"]
