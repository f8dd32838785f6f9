package gpu

/*
#include "ii2.h"
*/
import "C"

import "unsafe"

// This file is what the reference's `file` package needs to keep merged segments in the DV1
// layout instead of intcomp runs: the values file <key>_val holds the three DV1 arrays of a
// device segment (ExportSegment), and a reader hands them back to the device (ImportSegment) instead of
// calling intcomp.UncompressUint32 per term (file/reader.go:79-100).  The FST side of the
// package (term -> list index instead of term -> byte offset) stays Go.  The C++ host mirror
// (inverted_index_2_amd/host/segment_file.h) shows the same lifecycle end to end on its own
// file format, tested against TestWriter / TestWriterDirect / TestInitFromExistingFiles.
//
// Like ii2.go this has never been compiled here (no Go toolchain in the image).

// Skip is one ii2_skip entry: the first doc id of a DV1 block and the byte offset of its payload.
type Skip struct {
	FirstDoc uint32
	ByteOff  uint32
}

// DV1 is a segment's on-disk image: BlkOff[nLists+1], Skip[nBlocks+1], Payload[nBytes].
type DV1 struct {
	NPostings uint64
	BlkOff    []uint32
	Skip      []Skip
	Payload   []byte
}

// ExportSegment copies a device segment's DV1 arrays out (what Writer.Close would put into <key>_val).
func (c *Ctx) ExportSegment(s *Segment) (*DV1, error) {
	var info C.ii2_seg_info
	C.ii2_seg_get_info(s.h, &info)
	d := &DV1{
		NPostings: uint64(info.n_postings),
		BlkOff:    make([]uint32, uint64(info.n_lists)+1),
		Skip:      make([]Skip, uint64(info.n_blocks)+1),
		Payload:   make([]byte, uint64(info.n_bytes)),
	}
	var pl *C.uint8_t
	if len(d.Payload) > 0 {
		pl = (*C.uint8_t)(unsafe.Pointer(&d.Payload[0]))
	}
	if rc := C.ii2_seg_export(c.h, s.h, u32ptr(d.BlkOff), (*C.ii2_skip)(unsafe.Pointer(&d.Skip[0])), pl); rc != 0 {
		return nil, c.err("export", rc)
	}
	return d, nil
}

// ImportSegment adopts DV1 arrays read from a values file.  The library is told the arrays'
// lengths and checks the closing entries against them, then validates the structure on the
// device (monotone offsets, block fill rule, posting count) before any kernel walks it: a
// corrupt file is an error, never a fault.
func (c *Ctx) ImportSegment(d *DV1) (*Segment, error) {
	if len(d.BlkOff) == 0 || len(d.Skip) == 0 {
		return nil, c.err("import", C.int(C.II2_EINVAL))
	}
	var pl *C.uint8_t
	if len(d.Payload) > 0 {
		pl = (*C.uint8_t)(unsafe.Pointer(&d.Payload[0]))
	}
	var s *C.ii2_seg
	rc := C.ii2_seg_import(c.h, C.uint64_t(len(d.BlkOff)-1), C.uint64_t(d.NPostings), C.uint64_t(len(d.Skip)-1),
		C.uint64_t(len(d.Payload)), u32ptr(d.BlkOff), (*C.ii2_skip)(unsafe.Pointer(&d.Skip[0])), pl, C.II2_HOST, &s)
	if rc != 0 {
		return nil, c.err("import", rc)
	}
	return &Segment{s}, nil
}
