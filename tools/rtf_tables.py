#!/usr/bin/env python3
"""Tables over SEVERAL models from the records of tools/rtf_sweep.py, in the two layouts the reference's table maker writes
(tools/rtf/get-rtf-tables.py: "Results by Metric and Batch Size" -- chunk sizes as rows, models as columns, one table per batch
size -- and "Results by Model" -- chunk sizes as rows, batch sizes as columns).

  python tools/rtf_tables.py out.md sweep_a.jsonl sweep_b.jsonl ...

Each .jsonl starts with the header record of its run ({"model": ..., "precision": ..., "one_sequence_ms": ...}), followed by one
record per (chunk_size, batch_size)."""
import json
import sys

METRICS = [("vram", "Max VRAM Usage (MB)", "{:.2f}"), ("minutes_per_sec", "Minutes of Audio Processed per Second", "{:.2f}")]


def load(path):
    head, recs = None, []
    for line in open(path):
        d = json.loads(line)
        if "chunk_size" in d:
            recs.append(d)
        else:
            head = d
    return head, recs


def cell(text, width):
    return " " + text.center(width) + " "


def table(title, row_keys, col_keys, value, row_name="Chunk Size", col_fmt=str):
    width = {c: max(12, len(col_fmt(c))) for c in col_keys}
    w0 = max(12, max(len(str(r)) for r in row_keys))
    out = [f"### {title}", "",
           "|" + cell(row_name, w0) + "|" + "".join(cell(col_fmt(c), width[c]) + "|" for c in col_keys),
           "|" + "-" * (w0 + 2) + "|" + "".join("-" * (width[c] + 2) + "|" for c in col_keys)]
    for r in row_keys:
        out.append("|" + cell(str(r), w0) + "|" + "".join(cell(value(r, c), width[c]) + "|" for c in col_keys))
    return "\n".join(out) + "\n\n"


def main():
    out_path, paths = sys.argv[1], sys.argv[2:]
    data, heads = {}, {}
    for p in paths:
        head, recs = load(p)
        name = (head or {}).get("model", p)
        heads[name] = head or {}
        data[name] = {(r["chunk_size"], r["batch_size"]): r for r in recs}
    models = sorted(data)
    chunks = sorted({k[0] for m in data.values() for k in m})
    batches = sorted({k[1] for m in data.values() for k in m})
    doc = "# RTF Analysis Results\n\n"
    doc += "One MI355X, one synthetic 30-minute file, `tools/rtf_sweep.py` (`utils.longform.decode_windows`: encoder + CTC log-softmax + greedy tokens + " \
           "stitching inside the timing).  Models, precision and the same file as ONE sequence:\n\n"
    for m in models:
        h = heads[m]
        doc += f"* `{m}`: {h.get('precision', '?')}; one sequence {h.get('one_sequence_ms', '?')} ms = " \
               f"{h.get('one_sequence_audio_sec_per_sec', '?')} audio-sec/sec; merge_frames {h.get('merge_frames', 0)}\n"
    doc += "\n## Results by Metric and Batch Size\n\n"
    for key, title, fmt in METRICS:
        doc += f"### {title}\n\n"
        for b in batches:
            doc += table(f"{key} Data (Batch Size {b})", chunks, models,
                         lambda c, m, b=b, key=key, fmt=fmt: fmt.format(data[m][(c, b)][key]) if (c, b) in data[m] else "-")
    doc += "## Results by Model\n\n"
    for m in models:
        doc += f"### {m}\n\n"
        for key, title, fmt in METRICS + [("share_of_one_sequence", "share of the one-sequence rate", "{:.2f}")]:
            doc += table(f"{m} - {key}", chunks, batches,
                         lambda c, b, m=m, key=key, fmt=fmt: fmt.format(data[m][(c, b)][key]) if (c, b) in data[m] else "-",
                         col_fmt=lambda b: f"BS {b}")
    with open(out_path, "w") as f:
        f.write(doc)
    print("wrote", out_path, "models:", ", ".join(models))


if __name__ == "__main__":
    main()
