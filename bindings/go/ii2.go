// Package gpu binds the posting-list hot path of lezhnev74/inverted_index_2 to libii2_hip.so
// (include/ii2.h) through cgo.  It replaces bodies, not signatures: the reference's exported
// API (inverted_index.go, shard.go) keeps its shape; see index.go for the drop-in methods.
//
// This image has no Go toolchain, so the package has never been compiled here; the same
// entry points are exercised through ctypes by the repository's -m gpu tests.
//
// cgo rules honoured: only flat slices cross (no Go pointer to Go pointer), buffers are pinned
// for the duration of one call, the library retains nothing.
package gpu

/*
#cgo CFLAGS: -I${SRCDIR}/../../include
#cgo LDFLAGS: -L${SRCDIR}/../../inverted_index_2_amd -lii2_hip -Wl,-rpath,${SRCDIR}/../../inverted_index_2_amd
#include "ii2.h"
*/
import "C"

import (
	"fmt"
	"unsafe"
)

// Ctx is one GPU + one HIP stream.  Calls on a Ctx are serialised by the library; give every
// worker goroutine of InvertedIndex.Merge its own Ctx (inverted_index.go:83-103).  Segments and
// tombstone bitmaps belong to the device and may be shared by all Ctx of that device.
type Ctx struct{ h *C.ii2_ctx }

func NewCtx(device int) (*Ctx, error) {
	var h *C.ii2_ctx
	if rc := C.ii2_ctx_create(C.int(device), 0, &h); rc != 0 {
		return nil, fmt.Errorf("gpu: ctx: %s (%d)", C.GoString(C.ii2_last_error(nil)), int(rc))
	}
	return &Ctx{h}, nil
}

func (c *Ctx) Close()      { C.ii2_ctx_destroy(c.h) }
func (c *Ctx) Device() int { return int(C.ii2_ctx_device(c.h)) }

// Counters reports how often this context repeated a call on its second path because a bounded wait between workgroups of
// one launch ran out (merges on the packing path; two-list ANDs / segment encodes without a look-back).  Results are identical
// either way; a non-zero count on production hardware is worth a bug report.
func (c *Ctx) Counters() (mergeRepeats, lookbackRepeats uint64) {
	var out [2]C.uint64_t
	C.ii2_ctx_counters(c.h, &out[0], 2)
	return uint64(out[0]), uint64(out[1])
}

func (c *Ctx) err(what string, rc C.int) error {
	return fmt.Errorf("gpu: %s: %s (%d)", what, C.GoString(C.ii2_last_error(c.h)), int(rc))
}

func u32ptr(s []uint32) *C.uint32_t {
	if len(s) == 0 {
		return nil
	}
	return (*C.uint32_t)(unsafe.Pointer(&s[0]))
}
func u64ptr(s []uint64) *C.uint64_t {
	if len(s) == 0 {
		return nil
	}
	return (*C.uint64_t)(unsafe.Pointer(&s[0]))
}

// Segment is a device-resident DV1 segment (the encode step, file/writer.go:32-59).
type Segment struct{ h *C.ii2_seg }

// Encode uploads nLists posting lists given CSR-style (postOff[nLists+1] into values).
func (c *Ctx) Encode(postOff []uint64, values []uint32) (*Segment, error) {
	var s *C.ii2_seg
	if rc := C.ii2_seg_encode(c.h, C.uint64_t(len(postOff)-1), u64ptr(postOff), u32ptr(values), C.II2_HOST, &s); rc != 0 {
		return nil, c.err("encode", rc)
	}
	return &Segment{s}, nil
}

// Decode is the decode step (file/reader.go:79-100).
func (c *Ctx) Decode(s *Segment) (postOff []uint64, values []uint32, err error) {
	var info C.ii2_seg_info
	C.ii2_seg_get_info(s.h, &info)
	postOff = make([]uint64, uint64(info.n_lists)+1)
	values = make([]uint32, uint64(info.n_postings))
	if rc := C.ii2_seg_decode(c.h, s.h, u64ptr(postOff), u32ptr(values), C.II2_HOST); rc != 0 {
		return nil, nil, c.err("decode", rc)
	}
	return postOff, values, nil
}

func (s *Segment) Free() { C.ii2_seg_free(s.h); s.h = nil }

// MergeAligned replaces the loop of Shard.Merge (shard.go:163-212) for k term-aligned segments
// held in host memory.  segOff: k*(nTerms+1) offsets (per segment, into that segment's slice of
// values); segBase: k+1 starts of the segments' slices in values; removed: RemovedLists.Values().
// A term with outOff[t+1]==outOff[t] has no survivors and is dropped by the caller
// (shard.go:192-194); termsOut == 0 means "write no segment" (shard.go:219-225).
func (c *Ctx) MergeAligned(k int, nTerms uint64, segOff, segBase []uint64, values, removed []uint32) (outOff []uint64, outVals []uint32, termsOut uint64, err error) {
	outOff = make([]uint64, nTerms+1)
	outVals = make([]uint32, len(values)+1)
	var st C.ii2_merge_stats
	rc := C.ii2_merge_host(c.h, C.uint32_t(k), C.uint64_t(nTerms), u64ptr(segOff), u64ptr(segBase), u32ptr(values),
		u32ptr(removed), C.uint64_t(len(removed)), u64ptr(outOff), u32ptr(outVals), C.uint64_t(len(outVals)), &st)
	if rc != 0 {
		return nil, nil, 0, c.err("merge", rc)
	}
	return outOff, outVals[:st.n_out], uint64(st.n_terms_out), nil
}

// Alignment is the device-resident result of AlignTerms.
type Alignment struct {
	h      *C.ii2_align
	NUnion uint64
	K      int
}

// AlignTerms replaces the k-way term-dictionary walk of makeIterator (shard.go:253-278,
// bytes.Compare order of file/types.go:24-26) with one call: the k sorted dictionaries are
// given flat (termBytes, termOff[nAll+1], segFirst[k+1] = index in termOff of each
// dictionary's first term); the union dictionary and the per-segment mapping stay on the GPU.
func (c *Ctx) AlignTerms(termBytes []byte, termOff []uint64, segFirst []uint64) (*Alignment, error) {
	var tb *C.uint8_t
	if len(termBytes) > 0 {
		tb = (*C.uint8_t)(unsafe.Pointer(&termBytes[0]))
	}
	var a *C.ii2_align
	if rc := C.ii2_align_terms(c.h, C.uint32_t(len(segFirst)-1), tb, u64ptr(termOff), u64ptr(segFirst), &a); rc != 0 {
		return nil, c.err("align", rc)
	}
	var n C.uint64_t
	var k C.uint32_t
	C.ii2_align_info(a, &n, &k)
	return &Alignment{h: a, NUnion: uint64(n), K: int(k)}, nil
}

// Export copies the alignment out: rep[u] = index (into termOff) of one input term equal to union
// term u; srcList[s*NUnion+u] = index inside dictionary s of the term equal to union term u, or -1.
func (c *Ctx) Export(a *Alignment) (rep []uint64, srcList []int64, err error) {
	rep = make([]uint64, a.NUnion)
	srcList = make([]int64, uint64(a.K)*a.NUnion)
	var sl *C.int64_t
	if len(srcList) > 0 {
		sl = (*C.int64_t)(unsafe.Pointer(&srcList[0]))
	}
	if rc := C.ii2_align_export(c.h, a.h, u64ptr(rep), sl); rc != 0 {
		return nil, nil, c.err("align export", rc)
	}
	return rep, srcList, nil
}

// SelectAligned builds, on the GPU, the term-aligned view of a resident segment for dictionary s
// of the alignment (firstList = list of seg that corresponds to the dictionary's first term).
func (c *Ctx) SelectAligned(seg *Segment, a *Alignment, s int, firstList uint64) (*Segment, error) {
	var v *C.ii2_seg
	if rc := C.ii2_seg_select_aligned(c.h, seg.h, a.h, C.uint32_t(s), C.uint64_t(firstList), &v); rc != 0 {
		return nil, c.err("select", rc)
	}
	return &Segment{v}, nil
}

// SelectAlignedAll builds the views of all the alignment's dictionaries in one call (one wait instead of one per view).
func (c *Ctx) SelectAlignedAll(segs []*Segment, a *Alignment, firstList []uint64) ([]*Segment, error) {
	// the C side writes a.K views: the slices must be exactly that long
	if len(segs) == 0 || len(segs) != a.K || (len(firstList) != 0 && len(firstList) != a.K) {
		return nil, fmt.Errorf("select: %d segments / %d first lists for an alignment of %d dictionaries", len(segs), len(firstList), a.K)
	}
	hs := make([]*C.ii2_seg, len(segs))
	for i, s := range segs {
		hs[i] = s.h
	}
	outs := make([]*C.ii2_seg, len(segs))
	if rc := C.ii2_seg_select_aligned_all(c.h, (**C.ii2_seg)(unsafe.Pointer(&hs[0])), a.h, u64ptr(firstList), (**C.ii2_seg)(unsafe.Pointer(&outs[0]))); rc != 0 {
		return nil, c.err("select", rc)
	}
	vs := make([]*Segment, len(outs))
	for i, h := range outs {
		vs[i] = &Segment{h}
	}
	return vs, nil
}

func (a *Alignment) Free() { C.ii2_align_free(a.h); a.h = nil }

// Dictionary is a segment's sorted, duplicate-free term dictionary resident in HBM (ii2_dict): made once, when the
// segment is written or loaded, so that alignments read it in place (no upload per merge).
type Dictionary struct{ h *C.ii2_dict }

// NewDictionary uploads a dictionary given flat (termBytes, termOff[n+1], termOff[0] == 0).
func (c *Ctx) NewDictionary(termBytes []byte, termOff []uint64) (*Dictionary, error) {
	var tb *C.uint8_t
	if len(termBytes) > 0 {
		tb = (*C.uint8_t)(unsafe.Pointer(&termBytes[0]))
	}
	var d *C.ii2_dict
	if rc := C.ii2_dict_create(c.h, tb, u64ptr(termOff), C.uint64_t(len(termOff)-1), C.II2_HOST, &d); rc != 0 {
		return nil, c.err("dictionary", rc)
	}
	return &Dictionary{d}, nil
}

func (d *Dictionary) Free() { C.ii2_dict_free(d.h); d.h = nil }

// AlignDicts is AlignTerms on resident dictionaries (ii2_align_dicts): nothing is uploaded, nothing is sorted.
func (c *Ctx) AlignDicts(dicts []*Dictionary) (*Alignment, error) {
	hs := make([]*C.ii2_dict, len(dicts))
	for i, d := range dicts {
		hs[i] = d.h
	}
	var a *C.ii2_align
	if len(hs) == 0 {
		return nil, fmt.Errorf("align: no dictionaries")
	}
	if rc := C.ii2_align_dicts(c.h, C.uint32_t(len(hs)), (**C.ii2_dict)(unsafe.Pointer(&hs[0])), &a); rc != 0 {
		return nil, c.err("align", rc)
	}
	var n C.uint64_t
	var k C.uint32_t
	C.ii2_align_info(a, &n, &k)
	return &Alignment{h: a, NUnion: uint64(n), K: int(k)}, nil
}

// MergeSmall is the common Shard.Merge in one launch (ii2_merge_small): a few small resident segments (what Shard.Put
// writes, shard.go:33-67), their dictionaries flat as for AlignTerms, RemovedLists.Values().  It returns the merged
// segment (nil when no term survives, shard.go:219-225) and, per output list, the index into termOff of an input term
// equal to its term.  ErrTooLarge (II2_ERANGE) means: take AlignTerms + MergeSegmentsToSeg instead.
func (c *Ctx) MergeSmall(segs []*Segment, termBytes []byte, termOff, segFirst []uint64, removed []uint32) (*Segment, []uint64, error) {
	hs := make([]*C.ii2_seg, len(segs))
	for i, s := range segs {
		hs[i] = s.h
	}
	var tb *C.uint8_t
	if len(termBytes) > 0 {
		tb = (*C.uint8_t)(unsafe.Pointer(&termBytes[0]))
	}
	kept := make([]uint64, len(termOff))
	var nKept C.uint64_t
	var out *C.ii2_seg
	var st C.ii2_merge_stats
	if len(hs) == 0 {
		return nil, nil, fmt.Errorf("merge: no segments")
	}
	rc := C.ii2_merge_small(c.h, C.uint32_t(len(hs)), (**C.ii2_seg)(unsafe.Pointer(&hs[0])), tb, u64ptr(termOff), u64ptr(segFirst),
		u32ptr(removed), C.uint64_t(len(removed)), &out, u64ptr(kept), &nKept, &st)
	if rc == C.II2_ERANGE {
		return nil, nil, ErrTooLarge
	}
	if rc != 0 {
		return nil, nil, c.err("merge", rc)
	}
	if out == nil {
		return nil, nil, nil
	}
	return &Segment{out}, kept[:nKept], nil
}

// ReadSmall is the small Shard.Read in one launch (ii2_read_small): the merged lists of a few small resident segments —
// dictionary s names the lists listFirst[s].. of segs[s] (nil: from list 0) — on the host.  rep[j] indexes termOff: an
// input term equal to merged term j; the ids of term j are values[postOff[j]:postOff[j+1]].  ErrTooLarge as for MergeSmall.
func (c *Ctx) ReadSmall(segs []*Segment, termBytes []byte, termOff, segFirst, listFirst []uint64, capValues int) (rep, postOff []uint64, values []uint32, err error) {
	hs := make([]*C.ii2_seg, len(segs))
	for i, s := range segs {
		hs[i] = s.h
	}
	var tb *C.uint8_t
	if len(termBytes) > 0 {
		tb = (*C.uint8_t)(unsafe.Pointer(&termBytes[0]))
	}
	rep = make([]uint64, len(termOff))
	postOff = make([]uint64, len(termOff)+1)
	values = make([]uint32, capValues+1)
	var nUnion C.uint64_t
	if len(hs) == 0 {
		return nil, nil, nil, fmt.Errorf("read: no segments")
	}
	rc := C.ii2_read_small(c.h, C.uint32_t(len(hs)), (**C.ii2_seg)(unsafe.Pointer(&hs[0])), tb, u64ptr(termOff), u64ptr(segFirst),
		u64ptr(listFirst), u64ptr(rep), u64ptr(postOff), u32ptr(values), C.uint64_t(capValues), &nUnion)
	if rc == C.II2_ERANGE {
		return nil, nil, nil, ErrTooLarge
	}
	if rc != 0 {
		return nil, nil, nil, c.err("read", rc)
	}
	return rep[:nUnion], postOff[:nUnion+1], values[:postOff[nUnion]], nil
}

// ErrTooLarge: the inputs exceed the one-launch merge's limits (II2_SMALL_MERGE_*).
var ErrTooLarge = fmt.Errorf("gpu: too large for the one-launch merge")

// SegAllGather concatenates every rank's merged segment in rank order into one segment on every rank; the postings
// travel DV1-encoded (ii2_seg_allgather; one process per GPU, the communicator set up with ii2_comm_init).
func (c *Ctx) SegAllGather(local *Segment) (*Segment, error) {
	var out *C.ii2_seg
	if rc := C.ii2_seg_allgather(c.h, local.h, &out); rc != 0 {
		return nil, c.err("segment all-gather", rc)
	}
	return &Segment{out}, nil
}

// SegConcat is the same concatenation on one device: the lists of segs[0], then segs[1], ... as one segment.
func (c *Ctx) SegConcat(segs []*Segment) (*Segment, error) {
	if len(segs) == 0 {
		return nil, fmt.Errorf("concat: no segments")
	}
	hs := make([]*C.ii2_seg, len(segs))
	for i, s := range segs {
		hs[i] = s.h
	}
	var out *C.ii2_seg
	if rc := C.ii2_seg_concat(c.h, C.uint32_t(len(hs)), (**C.ii2_seg)(unsafe.Pointer(&hs[0])), &out); rc != 0 {
		return nil, c.err("concat", rc)
	}
	return &Segment{out}, nil
}

// MergeSegmentsToSeg merges k term-aligned resident segments (views from SelectAligned / SelectAlignedAll) into a new
// resident segment, minus RemovedLists.Values() (ii2_tomb_create + ii2_merge_segments_to_seg): the general path behind
// MergeSmall.  It returns nil when no posting survives (shard.go:219-225); the caller drops the emptied term slots.
func (c *Ctx) MergeSegmentsToSeg(segs []*Segment, removed []uint32) (*Segment, error) {
	hs := make([]*C.ii2_seg, len(segs))
	for i, s := range segs {
		hs[i] = s.h
	}
	var tomb *C.ii2_tomb
	if len(removed) > 0 {
		if rc := C.ii2_tomb_create(c.h, u32ptr(removed), C.uint64_t(len(removed)), C.II2_HOST, &tomb); rc != 0 {
			return nil, c.err("merge", rc)
		}
		defer C.ii2_tomb_free(tomb)
	}
	var out *C.ii2_seg
	var st C.ii2_merge_stats
	if len(hs) == 0 {
		return nil, fmt.Errorf("merge: no segments")
	}
	if rc := C.ii2_merge_segments_to_seg(c.h, C.uint32_t(len(hs)), (**C.ii2_seg)(unsafe.Pointer(&hs[0])), tomb, &out, &st); rc != 0 {
		return nil, c.err("merge", rc)
	}
	if out == nil {
		return nil, nil
	}
	return &Segment{out}, nil
}

// Union replaces PrefixSearch's append + slices.Sort + slices.Compact (inverted_index.go:274-292).
func (c *Ctx) Union(listOff []uint64, values, removed []uint32) ([]uint32, error) {
	return c.lists(true, listOff, values, removed)
}

// Intersect: ids present in every list (additive operator; the reference has none).
func (c *Ctx) Intersect(listOff []uint64, values, removed []uint32) ([]uint32, error) {
	return c.lists(false, listOff, values, removed)
}

func (c *Ctx) lists(union bool, listOff []uint64, values, removed []uint32) ([]uint32, error) {
	out := make([]uint32, len(values)+1)
	var n C.uint64_t
	var rc C.int
	if union {
		rc = C.ii2_union_host(c.h, C.uint32_t(len(listOff)-1), u64ptr(listOff), u32ptr(values), u32ptr(removed), C.uint64_t(len(removed)), u32ptr(out), C.uint64_t(len(out)), &n)
	} else {
		rc = C.ii2_intersect_host(c.h, C.uint32_t(len(listOff)-1), u64ptr(listOff), u32ptr(values), u32ptr(removed), C.uint64_t(len(removed)), u32ptr(out), C.uint64_t(len(out)), &n)
	}
	if rc != 0 {
		return nil, c.err("lists", rc)
	}
	return out[:n], nil
}
