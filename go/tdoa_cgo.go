// tdoa_cgo.go -- the cgo shim a maintainer of KX0U-Jim/tdoa-geolocation adds next to processor.go
// (as tdoa_gpu.go) to route the correlation path through libtdoa_mi355x.so.
//
// SOURCE ONLY: there is no Go toolchain in the build image, so this file is never compiled here.
// It is a pointer + length pass-through with no logic; everything it forwards to is exercised
// through the same C ABI by tests/ (ctypes) and examples/pair_from_c.c (C99).
// Call sites it replaces: processor.go:818, :838 (crossCorrelate), correlation_sanity.go:50,55;
// see INTEGRATION.md section 1 for the full table.
package main

/*
#cgo CFLAGS: -I${SRCDIR}/tdoa-mi355x/include
#cgo LDFLAGS: -L${SRCDIR}/tdoa-mi355x/tdoa-geolocation_amd -ltdoa_mi355x -Wl,-rpath,${SRCDIR}/tdoa-mi355x/tdoa-geolocation_amd
#include <stdlib.h>
#include "tdoa_mi355x.h"
*/
import "C"

import (
	"fmt"
	"unsafe"
)

// gpuCorrelator owns one tdoa_ctx (one GPU). Not safe for concurrent use (the reference is single-goroutine).
type gpuCorrelator struct{ ctx *C.tdoa_ctx }

// goLagSet: search the lags timeDomainCorrelation searches (processor.go:650-678: shorter input = template, lags
// [0, max(1, min(maxLag, Ls-Lt))), first strict maximum) instead of the signed range -maxLag < lag < maxLag.
func newGPUCorrelator(device int, goLagSet bool) (*gpuCorrelator, error) {
	var p C.tdoa_params
	C.tdoa_default_params(&p) // 2e6 Hz, maxLag 20000, block 1000, gate 0.001, window 2 000 000
	p.device = C.int32_t(device)
	if goLagSet {
		p.lag_mode = C.TDOA_LAGS_GO
	}
	var ctx *C.tdoa_ctx
	if rc := C.tdoa_create(&p, &ctx); rc != C.TDOA_OK {
		return nil, fmt.Errorf("tdoa_create: %s", C.GoString(C.tdoa_strerror(rc)))
	}
	return &gpuCorrelator{ctx: ctx}, nil
}

func (g *gpuCorrelator) Close() { C.tdoa_destroy(g.ctx) }

func c64ptr(s []complex64) *C.float {
	if len(s) == 0 {
		return nil
	}
	return (*C.float)(unsafe.Pointer(&s[0])) // Go complex64 == {float32 re, float32 im}
}

// crossCorrelate is the drop-in for (*TDOAProcessor).crossCorrelate (processor.go:619).
func (g *gpuCorrelator) crossCorrelate(signal1, signal2 []complex64) (int, float64) {
	var delay C.int32_t
	var corr C.double
	rc := C.tdoa_cross_correlate_c64(g.ctx, c64ptr(signal1), C.size_t(len(signal1)),
		c64ptr(signal2), C.size_t(len(signal2)), &delay, &corr)
	if rc != C.TDOA_OK {
		panic(fmt.Sprintf("tdoa_cross_correlate_c64: %s (%s)", C.GoString(C.tdoa_strerror(rc)),
			C.GoString(C.tdoa_last_error(g.ctx))))
	}
	return int(delay), float64(corr)
}

// processCaptures is the batched path: raw .dat bytes of every station in, one peak per
// (window, pair) out, pairs ordered i<j like processor.go:816-817.
func (g *gpuCorrelator) processCaptures(dat [][]byte) ([]C.tdoa_peak, int, error) {
	for s, b := range dat {
		if len(b) < 2 { // &b[0] of an empty slice panics; an empty capture cannot be windowed anyway
			return nil, 0, fmt.Errorf("upload %d: capture is empty", s)
		}
		rc := C.tdoa_capture_upload(g.ctx, C.int(s), (*C.uint8_t)(unsafe.Pointer(&b[0])), C.size_t(len(b)/2))
		if rc != C.TDOA_OK {
			return nil, 0, fmt.Errorf("upload %d: %s", s, C.GoString(C.tdoa_last_error(g.ctx)))
		}
	}
	var perBlock, total C.int
	C.tdoa_num_windows(g.ctx, &perBlock, &total)
	pairs := int(C.tdoa_num_pairs(g.ctx))
	out := make([]C.tdoa_peak, int(total)*pairs)
	if rc := C.tdoa_process(g.ctx, 0, 1, &out[0], nil); rc != C.TDOA_OK {
		return nil, 0, fmt.Errorf("tdoa_process: %s", C.GoString(C.tdoa_last_error(g.ctx)))
	}
	return out, pairs, nil
}

// processCapturesFine adds the sub-sample delay and the |TDOA| gate (in samples) to every peak.
func (g *gpuCorrelator) processCapturesFine(gate float64) ([]C.tdoa_peak, []C.tdoa_fine_peak, error) {
	var perBlock, total C.int
	C.tdoa_num_windows(g.ctx, &perBlock, &total)
	n := int(total) * int(C.tdoa_num_pairs(g.ctx))
	peaks, fine := make([]C.tdoa_peak, n), make([]C.tdoa_fine_peak, n)
	if rc := C.tdoa_process_fine(g.ctx, 0, 1, C.double(gate), &peaks[0], &fine[0]); rc != C.TDOA_OK {
		return nil, nil, fmt.Errorf("tdoa_process_fine: %s", C.GoString(C.tdoa_last_error(g.ctx)))
	}
	return peaks, fine, nil
}

// windowQuality is fastAnalyzeSamples' statistics (fast_analyzer.go:117-155) for every (window, station).
func (g *gpuCorrelator) windowQuality(stations int) ([]C.tdoa_window_quality, error) {
	var perBlock, total C.int
	C.tdoa_num_windows(g.ctx, &perBlock, &total)
	out := make([]C.tdoa_window_quality, int(total)*stations)
	if rc := C.tdoa_window_quality_all(g.ctx, 0, 1, &out[0]); rc != C.TDOA_OK {
		return nil, fmt.Errorf("tdoa_window_quality_all: %s", C.GoString(C.tdoa_last_error(g.ctx)))
	}
	return out, nil
}
