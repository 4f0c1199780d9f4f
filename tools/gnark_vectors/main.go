// Pin hand-off: prints, as JSON, values produced by the reference's real arithmetic (gnark-crypto v0.19.0, go.mod:5 of
// mmsyan/GoPairingBasedCryptography) at the boundary this repository replaces.  The build image has no Go toolchain, so
// this program was NOT run here; a maintainer with Go runs
//
//     cd tools/gnark_vectors && go mod init gnarkvectors && go get github.com/consensys/gnark-crypto@v0.19.0 && \
//         go run . > ../../tests/golden/gnark_vectors.json
//
// and tests/test_gnark_vectors.py then checks the oracle (CPU) and the engine (GPU) against every value — the single
// comparison that confirms or refutes what is otherwise "parity unpinned" (SURVEY.md §8c): the final-exponent cofactor,
// the tower basis order in memory and in GT.Bytes(), the Montgomery limb layout, Marshal()/Bytes() flag bits, the SVDW
// constants of HashToG1/G2 and G2 cofactor clearing.
//
// "raw" = the value's in-memory bytes (fp.Element = 4 little-endian uint64 Montgomery limbs), which is the C ABI layout.
package main

import (
	"encoding/hex"
	"encoding/json"
	"math/big"
	"os"
	"unsafe"

	"github.com/consensys/gnark-crypto/ecc/bn254"
	"github.com/consensys/gnark-crypto/ecc/bn254/fr"
)

func rawG1(p *bn254.G1Affine) string { return hex.EncodeToString(unsafe.Slice((*byte)(unsafe.Pointer(p)), 64)) }
func rawG2(p *bn254.G2Affine) string { return hex.EncodeToString(unsafe.Slice((*byte)(unsafe.Pointer(p)), 128)) }
func rawGT(p *bn254.GT) string       { return hex.EncodeToString(unsafe.Slice((*byte)(unsafe.Pointer(p)), 384)) }

func g1Entry(p *bn254.G1Affine) map[string]string {
	b := p.Bytes()
	return map[string]string{"raw": rawG1(p), "marshal": hex.EncodeToString(p.Marshal()), "bytes": hex.EncodeToString(b[:])}
}
func g2Entry(p *bn254.G2Affine) map[string]string {
	b := p.Bytes()
	return map[string]string{"raw": rawG2(p), "marshal": hex.EncodeToString(p.Marshal()), "bytes": hex.EncodeToString(b[:])}
}
func gtEntry(p *bn254.GT) map[string]string {
	b := p.Bytes()
	return map[string]string{"raw": rawGT(p), "bytes": hex.EncodeToString(b[:])}
}

func main() {
	_, _, g1, g2 := bn254.Generators()
	out := map[string]any{"gnark_crypto_version": "v0.19.0", "g1": g1Entry(&g1), "g2": g2Entry(&g2)}

	e, err := bn254.Pair([]bn254.G1Affine{g1}, []bn254.G2Affine{g2})
	if err != nil {
		panic(err)
	}
	out["pair_g1_g2"] = gtEntry(&e)

	// scalar multiplications by fixed scalars (decimal strings; the last one is r - 1)
	ks := []string{"1", "2", "3", "65537", "1311768467463790320",
		"6296462850587514219860166612309923493513421339716397012889107043265683424215",
		"21888242871839275222246405745257275088548364400416034343698204186575808495616"}
	var sm []map[string]any
	var pts1 []bn254.G1Affine
	var pts2 []bn254.G2Affine
	for _, s := range ks {
		k, _ := new(big.Int).SetString(s, 10)
		var p bn254.G1Affine
		var q bn254.G2Affine
		p.ScalarMultiplication(&g1, k)
		q.ScalarMultiplication(&g2, k)
		pts1, pts2 = append(pts1, p), append(pts2, q)
		sm = append(sm, map[string]any{"k": s, "g1": g1Entry(&p), "g2": g2Entry(&q)})
	}
	out["scalar_mul"] = sm

	// Pair on non-generator points, a 2-pair product, PairingCheck, GT.Exp / Mul / Div / Inverse
	e35, _ := bn254.Pair([]bn254.G1Affine{pts1[2]}, []bn254.G2Affine{pts2[3]}) // e([3]g1, [65537]g2)
	out["pair_3_65537"] = gtEntry(&e35)
	prod, _ := bn254.Pair([]bn254.G1Affine{pts1[1], pts1[4]}, []bn254.G2Affine{pts2[5], pts2[2]})
	out["pair_product_idx_1_5__4_2"] = gtEntry(&prod)
	var negG2 bn254.G2Affine
	negG2.Neg(&pts2[1])
	okCheck, _ := bn254.PairingCheck([]bn254.G1Affine{pts1[1], g1}, []bn254.G2Affine{g2, negG2}) // e([2]g1,g2) e(g1,-[2]g2) = 1
	out["pairing_check_true"] = okCheck
	kexp, _ := new(big.Int).SetString(ks[5], 10)
	var ex, mu, dv, iv bn254.GT
	ex.Exp(e, kexp)
	mu.Mul(&e, &e35)
	dv.Div(&e, &e35)
	iv.Inverse(&e35)
	out["gt_exp_pair_by_k5"] = gtEntry(&ex)
	out["gt_mul"] = gtEntry(&mu)
	out["gt_div"] = gtEntry(&dv)
	out["gt_inverse"] = gtEntry(&iv)

	// hash to curve with the four DSTs of hash/hash_to.go:114,170,205,272 and the demo DST of bls_signature_demo.go:25
	var h []map[string]any
	for _, msg := range []string{"abc", "", "GoPairingBasedCryptography"} {
		for _, dst := range []string{"Hash String To Element In G1", "Hash Bytes To Element In G1"} {
			p, err := bn254.HashToG1([]byte(msg), []byte(dst))
			if err != nil {
				panic(err)
			}
			h = append(h, map[string]any{"group": "g1", "msg": msg, "dst": dst, "point": g1Entry(&p)})
		}
		for _, dst := range []string{"Hash String To Element In G2", "Hash Bytes To Element In G2", "signature SigmaSignature"} {
			q, err := bn254.HashToG2([]byte(msg), []byte(dst))
			if err != nil {
				panic(err)
			}
			h = append(h, map[string]any{"group": "g2", "msg": msg, "dst": dst, "point": g2Entry(&q)})
		}
	}
	out["hash_to_curve"] = h

	// fr.Element layout (scalars cross the ABI as plain little-endian integers; this records gnark's own form for reference)
	var one fr.Element
	one.SetOne()
	out["fr_one_raw"] = hex.EncodeToString(unsafe.Slice((*byte)(unsafe.Pointer(&one)), 32))

	enc := json.NewEncoder(os.Stdout)
	enc.SetIndent("", " ")
	if err := enc.Encode(out); err != nil {
		panic(err)
	}
}
