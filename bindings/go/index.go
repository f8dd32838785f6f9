package gpu

// Drop-in bodies for the reference's exported methods.  Signatures are the reference's own,
// verbatim (inverted_index.go:41,62,113,192,300; shard.go:72,127); what changes is that the
// posting work goes through Ctx.  The segment files, the vellum FST walk and the locks stay
// the reference's Go code ("host" below = the unchanged parts of package inverted_index_2).
//
//	func (ii *InvertedIndex) Merge(reqCount, mCount, concurrency int) (mergedSegmentsLen int64, err error)
//	func (ii *InvertedIndex) Read(min, max []byte) (go_iterators.Iterator[file.TermValues], error)
//	func (ii *InvertedIndex) PrefixSearch(prefixes [][]byte) (found map[string][]uint32, err error)
//	func (s *Shard) Merge(reqCount, mCount int) (mergedSegmentsLen int, err error)
//	func (s *Shard) Read(min, max []byte) (go_iterators.Iterator[file.TermValues], error)

import (
	"errors"
	"fmt"
	"sync"
	"sync/atomic"
)

// TermValues mirrors file.TermValues (file/types.go:9-12).
type TermValues struct {
	Term   []byte
	Values []uint32
}

// EmptyIterator mirrors go_iterators.EmptyIterator: the sentinel Next returns at the end
// (shard.go:170, file/reader.go:41).
var EmptyIterator = errors.New("iterator is empty")

// Iterator mirrors go_iterators.Iterator[T].  Close MUST be called: it releases the segment
// read-locks taken by Shard.Read (shard.go:69-71,268-275).
type Iterator[T any] interface {
	Next() (T, error)
	Close() error
}

// HostShard is what the unchanged Go host code provides per shard: the term-aligned CSR of the
// segments a merge picked (FST walk, shard.go:127-158) and the writer for the result.
type HostShard interface {
	// PickAndLock picks <= mCount smallest segments with the merging CAS (shard.go:135-151) and
	// read-locks them; n < 2 means "nothing to merge".
	PickAndLock(reqCount, mCount int) (n int, err error)
	// Dictionaries returns the picked (or, for Read, all read-locked) segments' sorted term
	// dictionaries, flat, restricted to [min, max] (file/reader.go:33-71).
	Dictionaries(min, max []byte) (termBytes []byte, termOff, segFirst []uint64)
	// Postings returns segment s's decoded lists for the given source lists (-1 = empty slot).
	Postings(s int, srcList []int64) (off []uint64, values []uint32, err error)
	RemovedValues() []uint32 // removed_list.go:44-54
	// WriteMerged appends the surviving terms to a new segment and swaps it in (shard.go:197-242).
	WriteMerged(terms [][]byte, off []uint64, values []uint32) error
	Release() // readRelease / detach (shard.go:214,228-242)
}

// ShardMerge is the body of Shard.Merge (shard.go:127-245) with the loop :163-212 on the GPU.
//
// This body hands host slices over (MergeAligned = ii2_merge_host): the segments' postings cross PCIe on every call.  A shard
// that keeps its segments resident (Ctx.Encode once per segment, *Segment handles next to the file handles) replaces the
// whole body by ONE call for the usual merge of a few small segments - Ctx.MergeSmall (ii2_merge_small: alignment, union,
// removed-list filter, empty-term drop and encode in one launch, ~30 us) - and falls back to NewDictionary / AlignDicts /
// SelectAlignedAll / MergeSegmentsToSeg when it answers ErrTooLarge.  ShardRead likewise: Ctx.ReadSmall (ii2_read_small).
func ShardMerge(c *Ctx, s HostShard, reqCount, mCount int) (mergedSegmentsLen int, err error) {
	n, err := s.PickAndLock(reqCount, mCount)
	if err != nil || n < 2 {
		return 0, err
	}
	defer s.Release()
	tb, toff, first := s.Dictionaries(nil, nil)
	al, err := c.AlignTerms(tb, toff, first) // k-way dictionary merge on the device
	if err != nil {
		return 0, fmt.Errorf("s: merge: %w", err)
	}
	defer al.Free()
	rep, src, err := c.Export(al)
	if err != nil {
		return 0, fmt.Errorf("s: merge: %w", err)
	}
	nT := al.NUnion
	segOff := make([]uint64, 0, uint64(n)*(nT+1))
	segBase := make([]uint64, 1, n+1)
	var values []uint32
	for k := 0; k < n; k++ {
		off, v, perr := s.Postings(k, src[uint64(k)*nT:uint64(k+1)*nT])
		if perr != nil {
			return 0, fmt.Errorf("s: merge: %w", perr)
		}
		segOff = append(segOff, off...)
		values = append(values, v...)
		segBase = append(segBase, uint64(len(values)))
	}
	outOff, outVals, termsOut, err := c.MergeAligned(n, nT, segOff, segBase, values, s.RemovedValues())
	if err != nil {
		return 0, fmt.Errorf("s: merge: %w", err)
	}
	if termsOut > 0 { // lazy writer: nothing survives -> no segment (shard.go:219-225)
		terms := make([][]byte, 0, termsOut)
		off := make([]uint64, 1, termsOut+1)
		for t := uint64(0); t < nT; t++ {
			if outOff[t+1] > outOff[t] { // drop emptied terms (shard.go:192-194)
				terms = append(terms, tb[toff[rep[t]]:toff[rep[t]+1]])
				off = append(off, outOff[t+1])
			}
		}
		if err = s.WriteMerged(terms, off, outVals); err != nil {
			return 0, fmt.Errorf("s: merge: %w", err)
		}
	}
	return n, nil
}

// IndexMerge is the body of InvertedIndex.Merge (inverted_index.go:62-109): `concurrency` workers,
// one Ctx each, pull shards off one channel; a failing worker records the error and stops.
func IndexMerge(device int, shards []HostShard, reqCount, mCount, concurrency int) (mergedSegmentsLen int64, err error) {
	workCh := make(chan HostShard)
	go func() {
		for _, s := range shards {
			workCh <- s
		}
		close(workCh)
	}()
	var merged atomic.Int64
	var wg sync.WaitGroup
	for i := 0; i < concurrency; i++ {
		wg.Add(1)
		go func() {
			defer wg.Done()
			c, cerr := NewCtx(device)
			if cerr != nil {
				err = cerr
				return
			}
			defer c.Close()
			for s := range workCh {
				n, serr := ShardMerge(c, s, reqCount, mCount)
				if serr != nil {
					err = serr
					return
				}
				merged.Add(int64(n))
			}
		}()
	}
	wg.Wait()
	return merged.Load(), err
}

// readIterator serves Shard.Read / InvertedIndex.Read: the merged view is computed in one GPU
// call (no tombstones: the reference does not filter on read, shard.go:72-75) and then handed out
// term by term.  Next returns EmptyIterator at the end; Close releases the read-locks.
type readIterator struct {
	terms  [][]byte
	off    []uint64
	values []uint32
	pos    int
	host   HostShard
	closed bool
}

func (it *readIterator) Next() (tv TermValues, err error) {
	if it.pos >= len(it.terms) {
		return tv, EmptyIterator
	}
	tv = TermValues{Term: it.terms[it.pos], Values: it.values[it.off[it.pos]:it.off[it.pos+1]]}
	it.pos++
	return tv, nil
}

func (it *readIterator) Close() error {
	if !it.closed {
		it.closed = true
		it.host.Release()
	}
	return nil
}

// ShardRead is the body of Shard.Read (shard.go:72-75 + makeIterator :253-278); the caller has
// read-locked all segments (readLockAll, segments.go:32).
func ShardRead(c *Ctx, s HostShard, nSegs int, min, max []byte) (Iterator[TermValues], error) {
	tb, toff, first := s.Dictionaries(min, max)
	al, err := c.AlignTerms(tb, toff, first)
	if err != nil {
		s.Release()
		return nil, fmt.Errorf("index read: %w", err)
	}
	defer al.Free()
	rep, src, err := c.Export(al)
	if err != nil {
		s.Release()
		return nil, fmt.Errorf("index read: %w", err)
	}
	nT := al.NUnion
	var segOff, segBase []uint64
	var values []uint32
	segBase = append(segBase, 0)
	for k := 0; k < nSegs; k++ {
		off, v, perr := s.Postings(k, src[uint64(k)*nT:uint64(k+1)*nT])
		if perr != nil {
			s.Release()
			return nil, fmt.Errorf("index read: %w", perr)
		}
		segOff = append(segOff, off...)
		values = append(values, v...)
		segBase = append(segBase, uint64(len(values)))
	}
	outOff, outVals, _, err := c.MergeAligned(nSegs, nT, segOff, segBase, values, nil)
	if err != nil {
		s.Release()
		return nil, fmt.Errorf("index read: %w", err)
	}
	terms := make([][]byte, nT)
	for t := range terms {
		terms[t] = tb[toff[rep[t]]:toff[rep[t]+1]]
	}
	return &readIterator{terms: terms, off: outOff, values: outVals, host: s}, nil
}
