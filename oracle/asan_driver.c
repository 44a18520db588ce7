/* TEST INFRASTRUCTURE ONLY: runs the fp32 C restatement (rnnt_cpu.c) on one case read from a file, for the
 * sanitizer build (`make asan`: -fsanitize=address,undefined).  A standalone program rather than a shared library,
 * so that the sanitizer runtime is linked in and nothing has to be preloaded into an interpreter.
 *   input : int32 B, T, U1, V, blank, has_scale, want_grad, inplace; float logits[B*T*U1*V];
 *           int32 labels[B*max(U1-1,0)]; int32 in_len[B]; int32 lab_len[B]; float scale[B] (if has_scale)
 *   output: int32 status; float costs[B]; float grad[B*T*U1*V] (if want_grad)
 * Every buffer is allocated at its exact size, so a read or write one element beyond any of them is reported. */
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

int rnnt_oracle_cpu_loss_grad(const float* logits, const int32_t* labels, const int32_t* in_len, const int32_t* lab_len,
                              int B, int maxT, int maxU1, int V, int blank, const float* grad_scale, float* costs,
                              float* grad, int num_threads);

static void* read_exact(FILE* f, size_t bytes) {
  void* p = malloc(bytes ? bytes : 1);
  if (!p || (bytes && fread(p, 1, bytes, f) != bytes)) {
    fprintf(stderr, "asan_driver: short input\n");
    exit(3);
  }
  return p;
}

int main(int argc, char** argv) {
  if (argc != 3) {
    fprintf(stderr, "usage: %s in.bin out.bin\n", argv[0]);
    return 3;
  }
  FILE* f = fopen(argv[1], "rb");
  if (!f) return 3;
  int32_t h[8];
  if (fread(h, sizeof(int32_t), 8, f) != 8) return 3;
  const int B = h[0], T = h[1], U1 = h[2], V = h[3], blank = h[4], has_scale = h[5], want_grad = h[6], inplace = h[7];
  const size_t n = (size_t)B * T * U1 * V;
  float* logits = read_exact(f, n * sizeof(float));
  int32_t* labels = read_exact(f, (size_t)B * (U1 > 1 ? U1 - 1 : 0) * sizeof(int32_t));
  int32_t* in_len = read_exact(f, (size_t)B * sizeof(int32_t));
  int32_t* lab_len = read_exact(f, (size_t)B * sizeof(int32_t));
  float* scale = has_scale ? read_exact(f, (size_t)B * sizeof(float)) : NULL;
  fclose(f);
  float* costs = malloc((size_t)B * sizeof(float));
  float* grad = NULL;
  if (want_grad) grad = inplace ? logits : malloc(n * sizeof(float));
  const int rc = rnnt_oracle_cpu_loss_grad(logits, labels, in_len, lab_len, B, T, U1, V, blank, scale, costs, grad, 2);
  FILE* o = fopen(argv[2], "wb");
  if (!o) return 3;
  const int32_t status = rc;
  fwrite(&status, sizeof(int32_t), 1, o);
  fwrite(costs, sizeof(float), (size_t)B, o);
  if (want_grad && rc == 0) fwrite(grad, sizeof(float), n, o);
  fclose(o);
  if (grad && grad != logits) free(grad);
  free(costs); free(scale); free(lab_len); free(in_len); free(labels); free(logits);
  return 0;
}
