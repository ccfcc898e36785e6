"""FASTQ ingest -> device read batches, and SAM emit: the data formats either side of the hot path.

Mirrors, for exactly what the path needs:
    FastQReader            src/data/records/fastq/reader.rs:17-187 (single-line FASTQ; same validation and messages)
    SamData::from_alignment src/data/records/sam/mod.rs:223-245   (POS = ref_range.start + 1, CIGAR = states, AS:i = score)
    Display for SamData     src/data/records/sam/std_traits.rs:3-44 (tab-separated, optional fields appended)
"""
from __future__ import annotations

import io
from dataclasses import dataclass, field
from typing import BinaryIO, Iterator, List, Optional

from .alignment import AlignmentBatch, ReadBatch, SOME


class FastQError(ValueError):
    """std::io::ErrorKind::InvalidData of the reference's reader."""


@dataclass
class FastQ:
    header: str
    sequence: bytes
    quality: bytes


def _chop_line_break(b: bytes) -> bytes:
    if b.endswith(b"\n"):
        b = b[:-1]
        if b.endswith(b"\r"):
            b = b[:-1]
    return b


class FastQReader:
    """Iterator over single-line FASTQ records (reader.rs:86-187)."""

    def __init__(self, inner: BinaryIO):
        self._f = inner

    @staticmethod
    def from_readable(inner: BinaryIO) -> "FastQReader":
        buffered = inner if isinstance(inner, io.BufferedReader) else io.BufferedReader(inner)
        if not buffered.peek(1):
            raise FastQError("No FASTQ data was found!")
        return FastQReader(buffered)

    @staticmethod
    def from_path(path) -> "FastQReader":
        try:
            f = open(path, "rb")
        except OSError as e:
            raise OSError(f"Failed to open path: {path}: {e}") from e
        try:
            return FastQReader.from_readable(f)
        except FastQError as e:
            raise FastQError(f"Failed to read data at path: {path}: {e}") from e

    def __iter__(self) -> Iterator[FastQ]:
        return self

    def __next__(self) -> FastQ:
        line = self._f.readline()
        if not line:
            raise StopIteration
        if not line.startswith(b"@"):
            raise FastQError("Missing '@' symbol at header line beginning! Ensure that the FASTQ file is not multi-line.")
        header = _chop_line_break(line[1:])
        if not header:
            raise FastQError("Missing FASTQ header!")
        try:
            header_s = header.decode("utf-8")
        except UnicodeDecodeError as e:
            raise FastQError(str(e)) from e
        seq = _chop_line_break(self._f.readline())
        if not seq:
            raise FastQError(f"Missing FASTQ sequence! See header: {header_s}")
        plus = self._f.readline()
        if not plus.startswith(b"+"):
            raise FastQError(f"Missing '+' line! Ensure that the FASTQ file is not multi-line. See header: {header_s}")
        qual = _chop_line_break(self._f.readline())
        if len(qual) != len(seq):
            if not qual:
                raise FastQError(f"Missing FASTQ quality scores! See header: {header_s}")
            raise FastQError(f"Sequence and quality score length mismatch ({len(seq)} ≠ {len(qual)})! See: {header_s}")
        if any(c < 33 or c > 126 for c in qual):  # QualityScores: graphic ASCII
            raise FastQError(f"Invalid quality score byte! See: {header_s}")
        return FastQ(header_s, seq, qual)


def batch_from_fastq(records: List[FastQ], device: int = 0) -> ReadBatch:
    """Concatenates the sequences into one device-resident ragged (or fixed-length) batch."""
    return ReadBatch.from_sequences([r.sequence for r in records], device)


@dataclass
class SamData:
    qname: str
    flag: int
    rname: str
    pos: int
    mapq: int
    cigar: str
    rnext: str = "*"
    pnext: int = 0
    tlen: int = 0
    seq: bytes = b"*"
    qual: bytes = b"*"
    opt_fields: List[str] = field(default_factory=list)

    @staticmethod
    def from_alignment(aln: AlignmentBatch, i: int, qname: str, flag: int, rname: str, mapq: int, seq: bytes, qual: bytes) -> "SamData":
        """sam/mod.rs:223-245: both SAM and Alignment exclude clipped bases from positions; only the 1-based shift remains."""
        r = aln.records[i]
        return SamData(qname, flag, rname, int(r["ref_start"]) + 1, mapq, aln.cigar(i), "*", 0, 0, seq, qual, [f"AS:i:{int(r['score'])}"])

    @staticmethod
    def unmapped(qname: str, seq: bytes, qual: bytes) -> "SamData":
        return SamData(qname, 4, "*", 0, 0, "*", "*", 0, 0, seq, qual, [])

    def __str__(self) -> str:
        seq = self.seq.decode() if self.seq else "*"
        qual = self.qual.decode() if self.qual else "*"
        core = f"{self.qname}\t{self.flag}\t{self.rname}\t{self.pos}\t{self.mapq}\t{self.cigar}\t{self.rnext}\t{self.pnext}\t{self.tlen}\t{seq}\t{qual}"
        return core + ("\t" + "\t".join(self.opt_fields) if self.opt_fields else "")


def align_fastq_to_sam(records: List[FastQ], reference: bytes, rname: str, matrix, gap_open: int, gap_extend: int,
                       device: int = 0) -> List[SamData]:
    """FASTQ records -> `into_local_profile(..).sw_align_from_i8(SeqSrc::Reference(reference))` on the GPU -> SAM records."""
    from .alignment import SeqSrc, into_local_profile

    qnames = [r.header.split()[0] if r.header.split() else r.header for r in records]
    aln = into_local_profile(batch_from_fastq(records, device), matrix, gap_open, gap_extend, device).sw_align_from_i8(SeqSrc.Reference(reference))
    out = []
    for i, r in enumerate(records):
        if int(aln.status[i]) == SOME:
            out.append(SamData.from_alignment(aln, i, qnames[i], 0, rname, 255, r.sequence, r.quality))
        else:
            out.append(SamData.unmapped(qnames[i], r.sequence, r.quality))
    return out
