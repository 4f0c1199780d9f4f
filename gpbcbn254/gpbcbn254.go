// Package gpbcbn254 mirrors the gnark-crypto bn254 calls used by mmsyan/GoPairingBasedCryptography and runs them on
// MI355X GPUs through the C ABI of libgpbc_bn254.so (include/gpbc_bn254.h).
//
// gnark's in-memory structs are the ABI buffers: fp.Element is [4]uint64 Montgomery little-endian, so a
// []bn254.G1Affine is already the 64-byte-stride array the library reads (unsafe.SliceData) and results land directly
// in bn254.GT / G1Affine / G2Affine values.  There is one code path: no call falls back to gnark's CPU arithmetic.
//
// NOT COMPILED in the build image of this repository (no Go toolchain there); the same surface is compiled and tested
// as C++ (include/gpbc_bn254.hpp, tests/cpp/*.cpp) and as Python ctypes (gopairingbasedcryptography_amd/bn254.py).
// INTEGRATION.md maps every function to the reference call sites it replaces.
package gpbcbn254

/*
#cgo CFLAGS: -I${SRCDIR}/../include
#cgo LDFLAGS: -L${SRCDIR}/../gopairingbasedcryptography_amd -lgpbc_bn254 -Wl,-rpath,${SRCDIR}/../gopairingbasedcryptography_amd
#include "gpbc_bn254.h"
*/
import "C"

import (
	"errors"
	"math/big"
	"runtime"
	"sync"
	"unsafe"

	"github.com/consensys/gnark-crypto/ecc/bn254"
	"github.com/consensys/gnark-crypto/ecc/bn254/fp"
	"github.com/consensys/gnark-crypto/ecc/bn254/fr"
)

var errSizes = errors.New("invalid inputs sizes") // gnark's own error text for Pair / PairingCheck

// gpbc_last_error() is thread-local on the C side and a goroutine may move to another OS thread between two cgo calls, so
// every exported function pins its goroutine for the length of the call (`defer pin()()` first thing): the status of a
// library call and the message that belongs to it are then read on the same thread.
func pin() func() {
	runtime.LockOSThread()
	return runtime.UnlockOSThread
}

func status(rc C.int) error {
	if rc == 0 {
		return nil
	}
	return errors.New(C.GoString(C.gpbc_last_error()))
}

func must(rc C.int) {
	if rc != 0 {
		panic("gpbcbn254: " + C.GoString(C.gpbc_last_error())) // gnark's methods cannot fail: surface engine errors loudly
	}
}

// Init binds the process to the given HIP devices (nil or empty: every visible device).  Host-slice batch calls then
// shard their index range over all of them inside the library; results do not depend on the number of devices.
func Init(devices []int) error {
	defer pin()()
	if len(devices) == 0 {
		n := int(C.gpbc_device_count())
		if n <= 0 {
			return errors.New(C.GoString(C.gpbc_last_error()))
		}
		for i := 0; i < n; i++ {
			devices = append(devices, i)
		}
	}
	d := make([]C.int, len(devices))
	for i, v := range devices {
		d[i] = C.int(v)
	}
	return status(C.gpbc_init_devices(&d[0], C.int(len(d))))
}

// NumDevices is the number of GPUs bound by Init.
func NumDevices() int { return int(C.gpbc_num_devices()) }

// InitCollectives opens one RCCL communicator over the bound devices: the partial sums of G1ScalarMulSum / G2ScalarMulSum
// are then exchanged by ncclAllGather over xGMI instead of through the host.
func InitCollectives() error {
	defer pin()()
	return status(C.gpbc_comm_init_all())
}

// Shutdown releases the library's device memory and communicators.
func Shutdown() error {
	defer pin()()
	destroyGeneratorTables() // before the devices go: the tables live in their HBM
	return status(C.gpbc_shutdown())
}

// Pair replaces bn254.Pair (cpabe/bsw07/bsw07_cpabe.go:75,184; access/tree/access_tree_node.go:106,110,119;
// bibe/afp25_bibe/afp25_bibe.go:227,395,399,403; ...): product of pairings, one final exponentiation.
func Pair(P []bn254.G1Affine, Q []bn254.G2Affine) (bn254.GT, error) {
	defer pin()()
	var gt bn254.GT
	if len(P) == 0 || len(P) != len(Q) {
		return gt, errSizes
	}
	seg := [2]C.uint64_t{0, C.uint64_t(len(P))}
	rc := C.gpbc_multi_pair(unsafe.Pointer(unsafe.SliceData(P)), unsafe.Pointer(unsafe.SliceData(Q)),
		&seg[0], 1, unsafe.Pointer(&gt))
	return gt, status(rc)
}

// PairingCheck replaces bn254.PairingCheck (signature/bls01_signature/bls_signature.go:81).
func PairingCheck(P []bn254.G1Affine, Q []bn254.G2Affine) (bool, error) {
	defer pin()()
	if len(P) == 0 || len(P) != len(Q) {
		return false, errSizes
	}
	seg := [2]C.uint64_t{0, C.uint64_t(len(P))}
	var ok C.uint8_t
	rc := C.gpbc_pairing_check(unsafe.Pointer(unsafe.SliceData(P)), unsafe.Pointer(unsafe.SliceData(Q)), &seg[0], 1, &ok)
	return ok == 1, status(rc)
}

// PairBatch is the batched form the engine adds: out[i] = Pair([P[i]], [Q[i]]).
func PairBatch(P []bn254.G1Affine, Q []bn254.G2Affine) ([]bn254.GT, error) {
	defer pin()()
	if len(P) == 0 || len(P) != len(Q) {
		return nil, errSizes
	}
	out := make([]bn254.GT, len(P))
	rc := C.gpbc_pair_batch(unsafe.Pointer(unsafe.SliceData(P)), unsafe.Pointer(unsafe.SliceData(Q)),
		C.size_t(len(P)), unsafe.Pointer(unsafe.SliceData(out)))
	return out, status(rc)
}

// MultiPair: out[j] = Pair(P[segOff[j]:segOff[j+1]], Q[segOff[j]:segOff[j+1]]) — the products the reference assembles
// from single pairings, GT.Mul / Div / Exp (ibe/bb04_ibe/bb04_ibe.go:213-225, access/tree/access_tree_node.go:106-157,
// bibe/afp25_bibe/afp25_bibe.go:395-413) with one final exponentiation per segment.
func MultiPair(P []bn254.G1Affine, Q []bn254.G2Affine, segOff []uint64) ([]bn254.GT, error) {
	defer pin()()
	if len(segOff) < 2 || len(P) != len(Q) || segOff[len(segOff)-1] != uint64(len(P)) {
		return nil, errSizes
	}
	out := make([]bn254.GT, len(segOff)-1)
	var p, q unsafe.Pointer
	if len(P) > 0 {
		p, q = unsafe.Pointer(unsafe.SliceData(P)), unsafe.Pointer(unsafe.SliceData(Q))
	}
	rc := C.gpbc_multi_pair(p, q, (*C.uint64_t)(unsafe.SliceData(segOff)), C.size_t(len(out)), unsafe.Pointer(unsafe.SliceData(out)))
	return out, status(rc)
}

// PairFixedQ: out[j] = Pair(P[j*m:(j+1)*m], Q) for one shared list Q of m points (a BSW07 key against many ciphertexts:
// access/tree/access_tree_node.go:106-119 under cpabe/bsw07/bsw07_cpabe.go:172-195; gnark: PrecomputeLines).
func PairFixedQ(P []bn254.G1Affine, Q []bn254.G2Affine) ([]bn254.GT, error) {
	defer pin()()
	if len(Q) == 0 || len(P) == 0 || len(P)%len(Q) != 0 {
		return nil, errSizes
	}
	out := make([]bn254.GT, len(P)/len(Q))
	rc := C.gpbc_multi_pair_fixed_q(unsafe.Pointer(unsafe.SliceData(P)), unsafe.Pointer(unsafe.SliceData(Q)),
		C.size_t(len(Q)), C.size_t(len(out)), unsafe.Pointer(unsafe.SliceData(out)))
	return out, status(rc)
}

// scalarBytes: big.Int -> 32-byte little-endian, reduced mod r (call sites pass fr.Element.BigInt values or rand.Int(r)).
func scalarBytes(s *big.Int, dst *[32]byte) {
	var t big.Int
	t.Mod(s, fr.Modulus())
	b := t.Bytes() // big-endian, at most 32 bytes after the reduction
	for i := range b {
		dst[len(b)-1-i] = b[i]
	}
}

// G1ScalarMultiplication replaces new(bn254.G1Affine).ScalarMultiplication(a, s)
// (signature/bls01_signature/bls_signature.go:45; cpabe/bsw07/bsw07_cpabe.go:69,149,157,160).
func G1ScalarMultiplication(p, a *bn254.G1Affine, s *big.Int) *bn254.G1Affine {
	defer pin()()
	var k [32]byte
	scalarBytes(s, &k)
	var out bn254.G1Affine // a and p may alias
	must(C.gpbc_g1_scalar_mul_batch(unsafe.Pointer(a), 1, unsafe.Pointer(&k[0]), 1, unsafe.Pointer(&out)))
	*p = out
	return p
}

// Fixed-base window tables of the two generators (1 MB / 2 MB of HBM), built on the first ScalarMultiplicationBase call:
// 32 mixed additions per multiplication instead of the variable-base kernel's doublings.
// A mutex, not a sync.Once: a Once whose function panics (a call before Init, a failed table build) is spent for good and would leave
// nil handles behind for every later call.  The handles are set only when both tables exist, a failed attempt is retried by the
// next call, and Shutdown destroys them so that Init after Shutdown builds fresh ones.
var (
	genMu            sync.Mutex
	g1Table, g2Table *C.gpbc_fixed_base
)

func generatorTables() (*C.gpbc_fixed_base, *C.gpbc_fixed_base) {
	genMu.Lock()
	defer genMu.Unlock()
	if g1Table == nil || g2Table == nil {
		_, _, g1, g2 := bn254.Generators()
		var t1, t2 *C.gpbc_fixed_base
		must(C.gpbc_g1_fixed_base_create(unsafe.Pointer(&g1), 1, &t1))
		if rc := C.gpbc_g2_fixed_base_create(unsafe.Pointer(&g2), 1, &t2); rc != 0 {
			msg := C.GoString(C.gpbc_last_error())
			C.gpbc_fixed_base_destroy(t1)
			panic("gpbcbn254: " + msg)
		}
		g1Table, g2Table = t1, t2
	}
	return g1Table, g2Table
}

func destroyGeneratorTables() {
	genMu.Lock()
	defer genMu.Unlock()
	if g1Table != nil {
		C.gpbc_fixed_base_destroy(g1Table)
	}
	if g2Table != nil {
		C.gpbc_fixed_base_destroy(g2Table)
	}
	g1Table, g2Table = nil, nil
}

// G1ScalarMultiplicationBase replaces new(bn254.G1Affine).ScalarMultiplicationBase(s)
// (cpabe/bsw07/bsw07_cpabe.go:69; bibe/afp25_bibe/afp25_bibe.go:160).
func G1ScalarMultiplicationBase(p *bn254.G1Affine, s *big.Int) *bn254.G1Affine {
	defer pin()()
	t1, _ := generatorTables()
	var k [32]byte
	scalarBytes(s, &k)
	must(C.gpbc_fixed_base_msm(t1, unsafe.Pointer(&k[0]), 1, unsafe.Pointer(p)))
	return p
}

// G2ScalarMultiplication replaces new(bn254.G2Affine).ScalarMultiplication(a, s)
// (signature/bls01_signature/bls_signature.go:63; cpabe/bsw07/bsw07_cpabe.go:73,83,103-121).
func G2ScalarMultiplication(p, a *bn254.G2Affine, s *big.Int) *bn254.G2Affine {
	defer pin()()
	var k [32]byte
	scalarBytes(s, &k)
	var out bn254.G2Affine
	must(C.gpbc_g2_scalar_mul_batch(unsafe.Pointer(a), 1, unsafe.Pointer(&k[0]), 1, unsafe.Pointer(&out)))
	*p = out
	return p
}

// G2ScalarMultiplicationBase replaces new(bn254.G2Affine).ScalarMultiplicationBase(s)
// (cpabe/bsw07/bsw07_cpabe.go:73; bibe/afp25_bibe/afp25_bibe.go:164-165).
func G2ScalarMultiplicationBase(p *bn254.G2Affine, s *big.Int) *bn254.G2Affine {
	defer pin()()
	_, t2 := generatorTables()
	var k [32]byte
	scalarBytes(s, &k)
	must(C.gpbc_fixed_base_msm(t2, unsafe.Pointer(&k[0]), 1, unsafe.Pointer(p)))
	return p
}

func frBytes(s []fr.Element) [][32]byte {
	k := make([][32]byte, len(s))
	for i := range s {
		var b big.Int
		scalarBytes(s[i].BigInt(&b), &k[i])
	}
	return k
}

// G1ScalarMultiplicationBatch: out[i] = [s[i]] bases[i]  (len(bases) == 1 shares the base, e.g. ScalarMultiplicationBase).
func G1ScalarMultiplicationBatch(bases []bn254.G1Affine, s []fr.Element) ([]bn254.G1Affine, error) {
	defer pin()()
	if len(s) == 0 {
		return nil, nil
	}
	if len(bases) != 1 && len(bases) != len(s) {
		return nil, errSizes
	}
	k := frBytes(s)
	out := make([]bn254.G1Affine, len(s))
	rc := C.gpbc_g1_scalar_mul_batch(unsafe.Pointer(unsafe.SliceData(bases)), C.size_t(len(bases)),
		unsafe.Pointer(unsafe.SliceData(k)), C.size_t(len(s)), unsafe.Pointer(unsafe.SliceData(out)))
	return out, status(rc)
}

// G2ScalarMultiplicationBatch: the same over G2.
func G2ScalarMultiplicationBatch(bases []bn254.G2Affine, s []fr.Element) ([]bn254.G2Affine, error) {
	defer pin()()
	if len(s) == 0 {
		return nil, nil
	}
	if len(bases) != 1 && len(bases) != len(s) {
		return nil, errSizes
	}
	k := frBytes(s)
	out := make([]bn254.G2Affine, len(s))
	rc := C.gpbc_g2_scalar_mul_batch(unsafe.Pointer(unsafe.SliceData(bases)), C.size_t(len(bases)),
		unsafe.Pointer(unsafe.SliceData(k)), C.size_t(len(s)), unsafe.Pointer(unsafe.SliceData(out)))
	return out, status(rc)
}

// G1ScalarMulSum = sum_i [s[i]] bases[i]: the verifier's side of BLS aggregate verification with random linear
// combination (a loop of ScalarMultiplication + Add in the reference's style, gka/agka09/asbb.go:193-220), sharded over
// all bound GPUs with the partial sums combined inside the library.
func G1ScalarMulSum(bases []bn254.G1Affine, s []fr.Element) (bn254.G1Affine, error) {
	defer pin()()
	var out bn254.G1Affine
	if len(bases) != len(s) {
		return out, errSizes
	}
	if len(s) == 0 {
		return out, nil
	}
	k := frBytes(s)
	rc := C.gpbc_g1_scalar_mul_sum(unsafe.Pointer(unsafe.SliceData(bases)), unsafe.Pointer(unsafe.SliceData(k)), C.size_t(len(s)), unsafe.Pointer(&out))
	return out, status(rc)
}

// G2ScalarMulSum: the same over G2.
func G2ScalarMulSum(bases []bn254.G2Affine, s []fr.Element) (bn254.G2Affine, error) {
	defer pin()()
	var out bn254.G2Affine
	if len(bases) != len(s) {
		return out, errSizes
	}
	if len(s) == 0 {
		return out, nil
	}
	k := frBytes(s)
	rc := C.gpbc_g2_scalar_mul_sum(unsafe.Pointer(unsafe.SliceData(bases)), unsafe.Pointer(unsafe.SliceData(k)), C.size_t(len(s)), unsafe.Pointer(&out))
	return out, status(rc)
}

// gtExp256: z = x^e for one 32-byte little-endian exponent.
func gtExp256(x *bn254.GT, e *[32]byte) bn254.GT {
	var z bn254.GT
	must(C.gpbc_gt_exp_batch(unsafe.Pointer(x), unsafe.Pointer(&e[0]), 1, unsafe.Pointer(&z)))
	return z
}

// GTExp replaces new(bn254.GT).Exp(x, k) (access/tree/access_tree_node.go:123,156; bibe/afp25_bibe/afp25_bibe.go:258-259).
// Like gnark it takes ANY big.Int and does not reduce it (x need not lie in the order-r subgroup: the reference's
// SetRandom messages do not): a negative k inverts x first; an exponent wider than the kernel's 256 bits is evaluated in
// 256-bit digits, x^k = prod_i (x^(2^(256 i)))^(k_i).  The reference's call sites pass values below r: one digit.
func GTExp(z *bn254.GT, x bn254.GT, k *big.Int) *bn254.GT {
	defer pin()()
	var abs big.Int
	abs.Abs(k)
	base := x
	if k.Sign() < 0 {
		var inv bn254.GT
		must(C.gpbc_gt_inverse_batch(unsafe.Pointer(&x), 1, unsafe.Pointer(&inv)))
		base = inv
	}
	var acc bn254.GT
	acc.SetOne()
	mask := new(big.Int).Sub(new(big.Int).Lsh(big.NewInt(1), 256), big.NewInt(1))
	var two255, two [32]byte
	two255[31] = 0x80
	two[0] = 2
	for first := true; ; first = false {
		var digit big.Int
		digit.And(&abs, mask)
		var e [32]byte
		b := digit.Bytes() // at most 32 bytes
		for i := range b {
			e[len(b)-1-i] = b[i]
		}
		t := gtExp256(&base, &e)
		if first {
			acc = t
		} else {
			must(C.gpbc_gt_mul_batch(unsafe.Pointer(&acc), unsafe.Pointer(&t), 1, unsafe.Pointer(&acc)))
		}
		abs.Rsh(&abs, 256)
		if abs.Sign() == 0 {
			break
		}
		h := gtExp256(&base, &two255) // base^(2^256) = (base^(2^255))^2
		base = gtExp256(&h, &two)
	}
	*z = acc
	return z
}

// GTMul, GTDiv, GTInverse replace new(bn254.GT).Mul(x, y) / Div(x, y) / Inverse(x) with gnark's signatures
// (access/tree/access_tree_node.go:114,157; cpabe/bsw07/bsw07_cpabe.go:189-190): z may alias an operand.
func GTMul(z, x, y *bn254.GT) *bn254.GT {
	defer pin()()
	var out bn254.GT
	must(C.gpbc_gt_mul_batch(unsafe.Pointer(x), unsafe.Pointer(y), 1, unsafe.Pointer(&out)))
	*z = out
	return z
}
func GTDiv(z, x, y *bn254.GT) *bn254.GT {
	defer pin()()
	var out bn254.GT
	must(C.gpbc_gt_div_batch(unsafe.Pointer(x), unsafe.Pointer(y), 1, unsafe.Pointer(&out)))
	*z = out
	return z
}
func GTInverse(z, x *bn254.GT) *bn254.GT {
	defer pin()()
	var out bn254.GT
	must(C.gpbc_gt_inverse_batch(unsafe.Pointer(x), 1, unsafe.Pointer(&out)))
	*z = out
	return z
}

// The batched forms: one element per index.
func GTMulBatch(a, b []bn254.GT) ([]bn254.GT, error) {
	defer pin()()
	if len(a) != len(b) {
		return nil, errSizes
	}
	out := make([]bn254.GT, len(a))
	if len(a) == 0 {
		return out, nil
	}
	rc := C.gpbc_gt_mul_batch(unsafe.Pointer(unsafe.SliceData(a)), unsafe.Pointer(unsafe.SliceData(b)), C.size_t(len(a)), unsafe.Pointer(unsafe.SliceData(out)))
	return out, status(rc)
}
func GTDivBatch(a, b []bn254.GT) ([]bn254.GT, error) {
	defer pin()()
	if len(a) != len(b) {
		return nil, errSizes
	}
	out := make([]bn254.GT, len(a))
	if len(a) == 0 {
		return out, nil
	}
	rc := C.gpbc_gt_div_batch(unsafe.Pointer(unsafe.SliceData(a)), unsafe.Pointer(unsafe.SliceData(b)), C.size_t(len(a)), unsafe.Pointer(unsafe.SliceData(out)))
	return out, status(rc)
}

// MarshalG1Batch / UnmarshalG1Batch replace element.Marshal() / g1.Unmarshal(data)
// (serialization/serialization_curve.go:5-7,17-21) n elements per call.  Unlike the reference, the decode error is
// returned per element, not dropped.
func MarshalG1Batch(pts []bn254.G1Affine, compressed bool) []byte {
	defer pin()()
	w, c := 64, C.int(0)
	if compressed {
		w, c = 32, 1
	}
	out := make([]byte, w*len(pts))
	if len(pts) > 0 {
		must(C.gpbc_g1_marshal_batch(unsafe.Pointer(unsafe.SliceData(pts)), C.size_t(len(pts)), c, unsafe.Pointer(unsafe.SliceData(out))))
	}
	return out
}
func UnmarshalG1Batch(data []byte, elemBytes int) ([]bn254.G1Affine, []bool, error) {
	defer pin()()
	if elemBytes != 32 && elemBytes != 64 {
		return nil, nil, errors.New("G1 element size must be 32 or 64")
	}
	n := len(data) / elemBytes
	out, ok8, ok := make([]bn254.G1Affine, n), make([]uint8, n), make([]bool, n)
	if n == 0 {
		return out, ok, nil
	}
	rc := C.gpbc_g1_unmarshal_batch(unsafe.Pointer(unsafe.SliceData(data)), C.size_t(elemBytes), C.size_t(n),
		unsafe.Pointer(unsafe.SliceData(out)), (*C.uint8_t)(unsafe.SliceData(ok8)))
	for i := range ok8 {
		ok[i] = ok8[i] == 1
	}
	return out, ok, status(rc)
}
func MarshalG2Batch(pts []bn254.G2Affine, compressed bool) []byte {
	defer pin()()
	w, c := 128, C.int(0)
	if compressed {
		w, c = 64, 1
	}
	out := make([]byte, w*len(pts))
	if len(pts) > 0 {
		must(C.gpbc_g2_marshal_batch(unsafe.Pointer(unsafe.SliceData(pts)), C.size_t(len(pts)), c, unsafe.Pointer(unsafe.SliceData(out))))
	}
	return out
}
func UnmarshalG2Batch(data []byte, elemBytes int) ([]bn254.G2Affine, []bool, error) {
	defer pin()()
	if elemBytes != 64 && elemBytes != 128 {
		return nil, nil, errors.New("G2 element size must be 64 or 128")
	}
	n := len(data) / elemBytes
	out, ok8, ok := make([]bn254.G2Affine, n), make([]uint8, n), make([]bool, n)
	if n == 0 {
		return out, ok, nil
	}
	rc := C.gpbc_g2_unmarshal_batch(unsafe.Pointer(unsafe.SliceData(data)), C.size_t(elemBytes), C.size_t(n),
		unsafe.Pointer(unsafe.SliceData(out)), (*C.uint8_t)(unsafe.SliceData(ok8)))
	for i := range ok8 {
		ok[i] = ok8[i] == 1
	}
	return out, ok, status(rc)
}
func MarshalGTBatch(gt []bn254.GT) []byte { // GT.Marshal() = GT.Bytes() (hash/hash_from_gt.go:5-8)
	defer pin()()
	out := make([]byte, 384*len(gt))
	if len(gt) > 0 {
		must(C.gpbc_gt_marshal_batch(unsafe.Pointer(unsafe.SliceData(gt)), C.size_t(len(gt)), unsafe.Pointer(unsafe.SliceData(out))))
	}
	return out
}

// UnmarshalG1, UnmarshalG2, UnmarshalGT replace (*G1Affine).Unmarshal(buf) / (*G2Affine).Unmarshal / (*GT).Unmarshal
// (serialization/serialization_curve.go:17-33, whose callers drop the error).  Like gnark they accept the compressed and the
// uncompressed form by the flag bits of the first byte and return an error for a short buffer, a coordinate >= p, a point off
// the curve or outside the subgroup; the receiver is then left zero.
var errShort = errors.New("short buffer")
var errInvalidEncoding = errors.New("invalid point encoding")

func UnmarshalG1(p *bn254.G1Affine, buf []byte) error {
	defer pin()()
	if len(buf) < 32 {
		return errShort
	}
	size := 64
	if buf[0]&0xC0 != 0 { // mCompressedSmallest, mCompressedLargest or mCompressedInfinity
		size = 32
	} else if len(buf) < 64 {
		return errShort
	}
	var ok C.uint8_t
	if err := status(C.gpbc_g1_unmarshal_batch(unsafe.Pointer(unsafe.SliceData(buf)), C.size_t(size), 1, unsafe.Pointer(p), &ok)); err != nil {
		return err
	}
	if ok != 1 {
		return errInvalidEncoding
	}
	return nil
}
func UnmarshalG2(p *bn254.G2Affine, buf []byte) error {
	defer pin()()
	if len(buf) < 64 {
		return errShort
	}
	size := 128
	if buf[0]&0xC0 != 0 {
		size = 64
	} else if len(buf) < 128 {
		return errShort
	}
	var ok C.uint8_t
	if err := status(C.gpbc_g2_unmarshal_batch(unsafe.Pointer(unsafe.SliceData(buf)), C.size_t(size), 1, unsafe.Pointer(p), &ok)); err != nil {
		return err
	}
	if ok != 1 {
		return errInvalidEncoding
	}
	return nil
}
func UnmarshalGT(z *bn254.GT, buf []byte) error {
	defer pin()()
	if len(buf) < 384 {
		return errShort
	}
	var ok C.uint8_t
	if err := status(C.gpbc_gt_unmarshal_batch(unsafe.Pointer(unsafe.SliceData(buf)), 1, unsafe.Pointer(z), &ok)); err != nil {
		return err
	}
	if ok != 1 {
		return errInvalidEncoding
	}
	return nil
}

// HashToG1 and HashToG2 replace bn254.HashToG1(msg, dst) / bn254.HashToG2(msg, dst) with gnark's signatures
// (hash/hash_to.go:114,170,205,272 — one call per string in the reference).
func HashToG1(msg, dst []byte) (bn254.G1Affine, error) {
	out, err := HashToG1Batch([][]byte{msg}, dst)
	if err != nil {
		return bn254.G1Affine{}, err
	}
	return out[0], nil
}
func HashToG2(msg, dst []byte) (bn254.G2Affine, error) {
	out, err := HashToG2Batch([][]byte{msg}, dst)
	if err != nil {
		return bn254.G2Affine{}, err
	}
	return out[0], nil
}

// HashToG1Batch replaces bn254.HashToG1(msg, dst) (hash/hash_to.go:113-119,169-175) for a batch: the engine hashes
// (expand_message_xmd with SHA-256, reduction to two field elements), maps both, adds.
func HashToG1Batch(msgs [][]byte, dst []byte) ([]bn254.G1Affine, error) {
	defer pin()()
	out := make([]bn254.G1Affine, len(msgs))
	if len(msgs) == 0 {
		return out, nil
	}
	if len(dst) > 255 {
		return nil, errors.New("invalid domain size (>255 bytes)") // what gnark's ExpandMsgXmd answers
	}
	data, off := flatten(msgs)
	rc := C.gpbc_hash_to_g1(unsafe.Pointer(unsafe.SliceData(data)), (*C.uint64_t)(unsafe.SliceData(off)), C.size_t(len(msgs)),
		dstPointer(dst), C.size_t(len(dst)), unsafe.Pointer(unsafe.SliceData(out)))
	return out, status(rc)
}

// MapToG1Batch is the group part alone for callers that hold gnark's fp.Hash(msg, dst, 2) output already: u has two
// elements per point, out[i] = MapToCurve1(u[2i]) + MapToCurve1(u[2i+1]).
func MapToG1Batch(u []fp.Element) ([]bn254.G1Affine, error) {
	defer pin()()
	if len(u)%2 != 0 {
		return nil, errors.New("two field elements per point")
	}
	out := make([]bn254.G1Affine, len(u)/2)
	if len(out) == 0 {
		return out, nil
	}
	rc := C.gpbc_g1_map_to_curve_batch(unsafe.Pointer(unsafe.SliceData(u)), C.size_t(len(out)), unsafe.Pointer(unsafe.SliceData(out)))
	return out, status(rc)
}

// HashToG2Batch replaces bn254.HashToG2 (hash/hash_to.go:204-210,271-277): four base-field elements per message,
// E2 j = elements 2j (A0) and 2j+1 (A1); hashing, both maps, addition and cofactor clearing in one call.
func HashToG2Batch(msgs [][]byte, dst []byte) ([]bn254.G2Affine, error) {
	defer pin()()
	out := make([]bn254.G2Affine, len(msgs))
	if len(msgs) == 0 {
		return out, nil
	}
	if len(dst) > 255 {
		return nil, errors.New("invalid domain size (>255 bytes)") // what gnark's ExpandMsgXmd answers
	}
	data, off := flatten(msgs)
	rc := C.gpbc_hash_to_g2(unsafe.Pointer(unsafe.SliceData(data)), (*C.uint64_t)(unsafe.SliceData(off)), C.size_t(len(msgs)),
		dstPointer(dst), C.size_t(len(dst)), unsafe.Pointer(unsafe.SliceData(out)))
	return out, status(rc)
}

// dstPointer: an empty domain-separation tag is legal (the library accepts dst == NULL with dst_len == 0, and so does this shim:
// unsafe.SliceData of an empty slice may be nil).
func dstPointer(dst []byte) unsafe.Pointer {
	if len(dst) == 0 {
		return nil
	}
	return unsafe.Pointer(unsafe.SliceData(dst))
}

// flatten lays the messages back to back with their n+1 byte offsets (the layout of gpbc_hash_to_*).
func flatten(msgs [][]byte) ([]byte, []uint64) {
	off := make([]uint64, len(msgs)+1)
	total := 0
	for i, m := range msgs {
		total += len(m)
		off[i+1] = uint64(total)
	}
	data := make([]byte, 0, total+1)
	for _, m := range msgs {
		data = append(data, m...)
	}
	if len(data) == 0 {
		data = append(data, 0) // SliceData of an empty slice may be nil
	}
	return data, off
}
