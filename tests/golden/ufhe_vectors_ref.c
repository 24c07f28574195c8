/* Generator of tests/golden/ufhe_vectors.npz (run by tests/golden/make_ufhe_golden.py in the build container, where /root/reference exists): drives the
 * REFERENCE's radix-integer application (applications/multi-ciphertext-arith, compiled unchanged from where it lies) on the REFERENCE's own library for a fixed
 * list of inputs and prints what it decrypts -- add, sub, ReLU, the encrypted 16-entry LUT, comparison, cleartext LUT and multiplication of signed 8-bit integers.  Own code; the printed numbers are the
 * fixture the digit-parallel forms of this repository (mosfhet_hip_vec_*) are held to on the GPU. */
#include "ufhe.h"

ufhe_priv_keyset _priv_key;   /* integer.c declares them extern */
ufhe_context _ctx;

int main(int argc, char **argv) {
  const int count = argc > 1 ? atoi(argv[1]) : 24;
  _priv_key = ufhe_new_priv_keyset(SET0);
  _ctx = ufhe_setup_context(ufhe_new_public_keyset(_priv_key, SET0));
  ufhe_integer a = ufhe_new_integer(8, true, _ctx), b = ufhe_new_integer(8, true, _ctx), c = ufhe_new_integer(8, true, _ctx), sel = ufhe_new_integer(4, true, _ctx);
  ufhe_integer wide = ufhe_new_integer(32, true, _ctx);
  ufhe_integer vec[16];
  for (int j = 0; j < 16; j++) vec[j] = ufhe_new_integer(8, true, _ctx);
  printf("{\"torus_base\": %d, \"rows\": [\n", _ctx->torus_base);
  for (int i = 0; i < count; i++) {
    const int8_t va = (int8_t)((37 * i + 11) & 0xff), vb = (int8_t)((91 * i + 5) & 0xff);
    const int vs = (5 * i + 2) & 0xf;
    ufhe_encrypt_integer(a, (uint64_t)(uint8_t)va, _priv_key, _ctx);
    ufhe_encrypt_integer(b, (uint64_t)(uint8_t)vb, _priv_key, _ctx);
    ufhe_add_integer(c, a, b, _ctx);
    const int r_add = (int)(int8_t)ufhe_decrypt_integer(c, _priv_key, _ctx);
    ufhe_sub_integer(c, a, b, _ctx);
    const int r_sub = (int)(int8_t)ufhe_decrypt_integer(c, _priv_key, _ctx);
    ufhe_relu_integer(c, a, _ctx);
    const int r_relu = (int)(int8_t)ufhe_decrypt_integer(c, _priv_key, _ctx);
    ufhe_encrypt_integer(sel, (uint64_t)vs, _priv_key, _ctx);
    printf("  {\"a\": %d, \"b\": %d, \"sel\": %d, \"table\": [", (int)va, (int)vb, vs);
    for (int j = 0; j < 16; j++) {
      const int8_t t = (int8_t)((13 * i + 7 * j * j + 3) & 0xff);
      ufhe_encrypt_integer(vec[j], (uint64_t)(uint8_t)t, _priv_key, _ctx);
      printf("%d%s", (int)t, j == 15 ? "" : ", ");
    }
    ufhe_mux_integer_array(c, sel, _ctx, 16, vec);
    const int r_lut = (int)(int8_t)ufhe_decrypt_integer(c, _priv_key, _ctx);
    /* comparison, signed and unsigned reading of the same bytes (test_int_cmp), and the cleartext-table look-up of test_lut_ct with the row's table as cleartext */
    ufhe_cmp_integer(c, a, b, _ctx);
    const int r_cmp_s = (int)ufhe_decrypt_integer(c, _priv_key, _ctx);
    a->_signed = b->_signed = c->_signed = false;
    ufhe_cmp_integer(c, a, b, _ctx);
    const int r_cmp_u = (int)ufhe_decrypt_integer(c, _priv_key, _ctx);
    a->_signed = b->_signed = c->_signed = true;
    uint64_t ct_lut[16];
    for (int j = 0; j < 16; j++) ct_lut[j] = (uint64_t)((13 * i + 7 * j * j + 3) & 0xff);
    ufhe_lut_integer(c, sel, ct_lut, 16, _ctx);
    const int r_lut_ct = (int)(int8_t)ufhe_decrypt_integer(c, _priv_key, _ctx);
    /* multiplication as test_int_mul has it: signed 8-bit operands into a 32-bit result (the low byte of the product, sign-extended), then the same bytes read unsigned
     * (the full product) */
    ufhe_mul_integer(wide, a, b, _ctx);
    const long long r_mul_s = (long long)(int32_t)ufhe_decrypt_integer(wide, _priv_key, _ctx);
    a->_signed = b->_signed = wide->_signed = false;
    ufhe_mul_integer(wide, a, b, _ctx);
    const long long r_mul_u = (long long)(uint32_t)ufhe_decrypt_integer(wide, _priv_key, _ctx);
    a->_signed = b->_signed = wide->_signed = true;
    /* shifted addition (test_int_sl_add): c = a B^g + b B^h with the row's shifts, signed (a's four digits, then the sign extension) and unsigned */
    const int sg = i & 3, sh = (i >> 2) & 3;
    ufhe_sl_add_integer(wide, a, sg, b, sh, _ctx);
    const long long r_sl_s = (long long)(int32_t)ufhe_decrypt_integer(wide, _priv_key, _ctx);
    a->_signed = b->_signed = wide->_signed = false;
    ufhe_sl_add_integer(wide, a, sg, b, sh, _ctx);
    const long long r_sl_u = (long long)(uint32_t)ufhe_decrypt_integer(wide, _priv_key, _ctx);
    a->_signed = b->_signed = wide->_signed = true;
    printf("], \"add\": %d, \"sub\": %d, \"relu\": %d, \"lut\": %d, \"cmp_signed\": %d, \"cmp_unsigned\": %d, \"lut_cleartext\": %d, \"mul_signed\": %lld, \"mul_unsigned\": %lld, "
           "\"sl_g\": %d, \"sl_h\": %d, \"sl_add_signed\": %lld, \"sl_add_unsigned\": %lld}%s\n",
           r_add, r_sub, r_relu, r_lut, r_cmp_s, r_cmp_u, r_lut_ct, r_mul_s, r_mul_u, sg, sh, r_sl_s, r_sl_u, i == count - 1 ? "" : ",");
    fflush(stdout);
  }
  printf("]}\n");
  return 0;
}
