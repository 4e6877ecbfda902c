// dump_golden_test.go — pins SURVEY rows a1-a10, a15-a17 against the REAL reference.
//
// THIS FILE HAS NEVER BEEN COMPILED: the image this repository is built in has no Go toolchain and the reference's
// modules (go-dsp, digimodes) are not vendored there.  It is written for a maintainer who has both: copy this
// directory into a checkout of github.com/ftl/sdrainer (say as ./integration/hipgolden), so that the imports below
// resolve inside the reference's module, and run
//
//	go test ./integration/hipgolden -run TestDumpGolden -v
//
// It reads the committed little-endian float32 IQ fixtures (testdata/*.f32, written by
// tests/golden/make_go_fixtures.py of the MI355X repository from the same seeded generator the GPU parity tests use),
// runs the reference's own code on them - dsp.FFT.IQToSpectrumAndPSD with the projection of rx/receiver.go:376-378,
// dsp.FindNoiseFloor, the threshold lines of rx/receiver.go:381-385, the cumulation of :404-407 and dsp.FindPeaks -
// and compares sha256 digests of what it gets with testdata/expected.json, which holds what the MI355X repository's
// CPU oracle (oracle/sdr_oracle.c, a C restatement of these very loops) and its HIP path produce for the same input.
// A mismatch names the first array that differs; the oracle is then wrong about go-dsp's factor table or about
// math.Log10 / math.Sincos, and so is the GPU path that is bit-identical to it.
//
// Known soft spot (DESIGN.md section 2): the restatement follows the pure-Go math.Log (FreeBSD e_log.c) and math.Sincos
// (Cephes).  As recalled - it could not be checked in the image this was written in - the toolchain the reference's go.mod
// names (go 1.23) uses those files on amd64 and arm64 (older toolchains had an assembly Log on amd64; s390x still has
// one).  If only spectrum_sha256 / nf_in / dev_in differ while psd_sha256 agrees, math.Log is where to look; if
// psd_sha256 differs, go-dsp's factor table (math.Sincos for the odd entries) is.
package hipgolden

import (
	"crypto/sha256"
	"encoding/binary"
	"encoding/hex"
	"encoding/json"
	"math"
	"os"
	"path/filepath"
	"testing"

	"github.com/ftl/sdrainer/dsp"
)

const (
	dBmShift       = 120 // rx/receiver.go:26
	cumulationSize = 100 // rx/receiver.go:22
	peakThreshold  = 15  // rx/receiver.go:24
)

type expectedCase struct {
	File         string            `json:"file"`
	SampleRate   int               `json:"sample_rate"`
	BlockSize    int               `json:"block_size"`
	Frames       int               `json:"frames"`
	EdgeWidth    int               `json:"edge_width"`
	CenterFreq   int               `json:"center_frequency"`
	Spectrum     string            `json:"spectrum_sha256"`
	PSD          string            `json:"psd_sha256"`
	Records      map[string]string `json:"records_sha256"`
	Cumulations  []string          `json:"cumulation_sha256"`
	Peaks        [][][7]float64    `json:"peaks"` // from, to, fromFrequency, toFrequency, signalFrequency, signalValue, signalBin
}

func digestF32(v []float32) string {
	buf := make([]byte, 4*len(v))
	for i, x := range v {
		binary.LittleEndian.PutUint32(buf[4*i:], math.Float32bits(x))
	}
	s := sha256.Sum256(buf)
	return hex.EncodeToString(s[:])
}

func digestF64(v []float64) string {
	buf := make([]byte, 8*len(v))
	for i, x := range v {
		binary.LittleEndian.PutUint64(buf[8*i:], math.Float64bits(x))
	}
	s := sha256.Sum256(buf)
	return hex.EncodeToString(s[:])
}

func readF32(t *testing.T, path string, n int) []float32 {
	raw, err := os.ReadFile(path)
	if err != nil {
		t.Fatal(err)
	}
	if len(raw) != 4*n {
		t.Fatalf("%s: %d bytes, want %d", path, len(raw), 4*n)
	}
	out := make([]float32, n)
	for i := range out {
		out[i] = math.Float32frombits(binary.LittleEndian.Uint32(raw[4*i:]))
	}
	return out
}

func TestDumpGolden(t *testing.T) {
	raw, err := os.ReadFile(filepath.Join("testdata", "expected.json"))
	if err != nil {
		t.Fatal(err)
	}
	var cases map[string]expectedCase
	if err := json.Unmarshal(raw, &cases); err != nil {
		t.Fatal(err)
	}
	for name, c := range cases {
		t.Run(name, func(t *testing.T) {
			n := c.BlockSize
			iq := readF32(t, filepath.Join("testdata", c.File), c.Frames*2*n)
			fft := dsp.NewFFT[float32]()
			mapping := dsp.NewFrequencyMapping[int](c.SampleRate, n, c.CenterFreq)
			noiseFloorMean := dsp.NewRollingMean[float32](60)    // rx/receiver.go:343
			noiseDeviationMean := dsp.NewRollingMean[float32](60) // :344
			spectrum := make(dsp.Block[float32], n)
			psd := make(dsp.Block[float32], n)
			cumulation := make(dsp.Block[float32], n)
			var peaks []dsp.Peak[float32, int]
			allSpectrum := make([]float32, 0, c.Frames*n)
			allPSD := make([]float32, 0, c.Frames*n)
			rec := map[string][]float32{}
			variance := make([]float64, 0, c.Frames)
			var cumDigests []string
			var peakLists [][][7]float64
			count := 0
			for f := 0; f < c.Frames; f++ {
				frame := iq[f*2*n : (f+1)*2*n]
				// rx/receiver.go:376-378
				fft.IQToSpectrumAndPSD(spectrum, psd, frame, func(v complex128, blockSize int) float32 {
					return dsp.MagnitudeIndB[float32](v, blockSize) + dBmShift
				})
				allSpectrum = append(allSpectrum, spectrum...)
				allPSD = append(allPSD, psd...)
				// :381-385
				psdNoiseFloor, noiseVariance := dsp.FindNoiseFloor(psd, c.EdgeWidth)
				devIn := float32(float64(dsp.PSDValueIndB(float32(math.Sqrt(noiseVariance)), n)+dBmShift) * 0.25)
				noiseDeviation := noiseDeviationMean.Put(devIn)
				nfIn := dsp.PSDValueIndB(psdNoiseFloor, n) + dBmShift
				noiseFloor := noiseFloorMean.Put(nfIn)
				threshold := float32(peakThreshold) + noiseFloor
				rec["min_mean"] = append(rec["min_mean"], psdNoiseFloor)
				variance = append(variance, noiseVariance)
				rec["dev_in"] = append(rec["dev_in"], devIn)
				rec["nf_in"] = append(rec["nf_in"], nfIn)
				rec["noise_dev"] = append(rec["noise_dev"], noiseDeviation)
				rec["noise_floor"] = append(rec["noise_floor"], noiseFloor)
				rec["peak_thr"] = append(rec["peak_thr"], threshold)
				rec["listen_thr"] = append(rec["listen_thr"], noiseFloor+noiseDeviation) // :394
				// :404-407, :459-460
				for i := range cumulation {
					cumulation[i] += spectrum[i]
				}
				count++
				if count == cumulationSize {
					cumDigests = append(cumDigests, digestF32(cumulation))
					peaks = dsp.FindPeaks(peaks, cumulation, cumulationSize, threshold, mapping)
					list := make([][7]float64, 0, len(peaks))
					for _, p := range peaks {
						list = append(list, [7]float64{float64(p.From), float64(p.To), float64(p.FromFrequency), float64(p.ToFrequency),
							float64(p.SignalFrequency), float64(p.SignalValue), float64(p.SignalBin)})
					}
					peakLists = append(peakLists, list)
					clear(cumulation)
					count = 0
				}
			}
			check := func(what, got, want string) {
				if got != want {
					t.Errorf("%s: %s differs: %s, the oracle has %s", name, what, got, want)
				}
			}
			check("psd", digestF32(allPSD), c.PSD)
			check("spectrum", digestF32(allSpectrum), c.Spectrum)
			check("variance", digestF64(variance), c.Records["variance"])
			for k, v := range rec {
				check(k, digestF32(v), c.Records[k])
			}
			if len(cumDigests) != len(c.Cumulations) {
				t.Fatalf("%d cumulations, the oracle has %d", len(cumDigests), len(c.Cumulations))
			}
			for i := range cumDigests {
				check("cumulation", cumDigests[i], c.Cumulations[i])
				if len(peakLists[i]) != len(c.Peaks[i]) {
					t.Errorf("cumulation %d: %d peaks, the oracle has %d", i, len(peakLists[i]), len(c.Peaks[i]))
					continue
				}
				for k := range peakLists[i] {
					g, w := peakLists[i][k], c.Peaks[i][k]
					for j := range g {
						if j == 5 { // signal value: float32, compare its bits
							if math.Float32bits(float32(g[j])) != math.Float32bits(float32(w[j])) {
								t.Errorf("cumulation %d peak %d: value %v, the oracle has %v", i, k, g[j], w[j])
							}
						} else if g[j] != w[j] {
							t.Errorf("cumulation %d peak %d field %d: %v, the oracle has %v", i, k, j, g[j], w[j])
						}
					}
				}
			}
		})
	}
}
