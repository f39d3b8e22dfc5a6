#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <vector>
#include <string>
#include "../include/os2s.h"
int main() {
  const char* G = "tests/golden/";
  std::string lm = std::string(G) + "ctc_test_lm.binary", trie = std::string(G) + "ctc_test_lm.trie";
  // alphabet file
  FILE* f = fopen("/tmp/alpha.txt", "w");
  const char* letters = " abcdefghijklmnopqrstuvwxyz'";
  std::vector<std::string> voc;
  for (const char* p = letters; *p; ++p) { fprintf(f, "%c\n", *p); voc.push_back(std::string(1, *p)); }
  fclose(f);
  void* sc = nullptr;
  int rc = os2s_ctc_scorer_create(lm.c_str(), trie.c_str(), "/tmp/alpha.txt", 2.0f, 0.5f, 0.1f, &sc);
  printf("scorer rc %d\n", rc);
  std::vector<const char*> vp; for (auto& v : voc) vp.push_back(v.c_str());
  void* ds = nullptr;
  rc = os2s_ctc_dict_scorer_create(lm.c_str(), vp.data(), (int)vp.size(), 2.0, 0.5, &ds);
  printf("dict scorer rc %d\n", rc);
  const int T = 120, B = 6, C = 29;
  std::vector<float> logits((size_t)T * B * C), probs((size_t)T * B * C);
  srand(1);
  for (int t = 0; t < T; ++t) for (int b = 0; b < B; ++b) {
    float sum = 0.f; float* l = &logits[((size_t)t * B + b) * C];
    for (int c = 0; c < C; ++c) l[c] = 6.f * rand() / RAND_MAX;
    // spell "ten seconds " repeatedly with some noise
    const char* w = "ten seconds "; int pos = (t / 3) % 12; int lab = w[pos] == ' ' ? 0 : (w[pos] - 'a' + 1);
    if (t % 3 == 2) lab = 28;
    l[lab] += 7.f;
    for (int c = 0; c < C; ++c) sum += std::exp(l[c]);
    for (int c = 0; c < C; ++c) probs[((size_t)t * B + b) * C + c] = std::exp(l[c]) / sum;
  }
  std::vector<int32_t> sl = {T, T - 5, 0, 1, 60, T};
  for (int beam : {1, 4, 64}) {
    const int top = beam >= 4 ? 2 : 1;
    std::vector<int32_t> ids((size_t)B * top * T), len(B * top); std::vector<float> lp(B * top);
    rc = os2s_ctc_beam_search(logits.data(), B * C, C, sl.data(), T, B, C, beam, 1, 0, sc, 3, ids.data(), len.data(), lp.data());
    printf("beam %d trie-decoder rc %d len0 %d lp0 %f\n", beam, rc, len[0], lp[0]);
    rc = os2s_ctc_beam_search(logits.data(), B * C, C, sl.data(), T, B, C, beam, 1, 1, nullptr, 2, ids.data(), len.data(), lp.data());
    printf("beam %d plain rc %d len0 %d\n", beam, rc, len[0]);
    rc = os2s_ctc_dict_beam_search(probs.data(), B * C, C, sl.data(), T, B, C, beam, 1.0, 40, 1, ds, 3, ids.data(), len.data(), lp.data());
    printf("beam %d dict-decoder rc %d len0 %d score0 %f\n", beam, rc, len[0], lp[0]);
    rc = os2s_ctc_dict_beam_search(probs.data(), B * C, C, sl.data(), T, B, C, beam, 0.98, 5, 1, nullptr, 1, ids.data(), len.data(), lp.data());
    printf("beam %d dict-decoder (no scorer, pruned) rc %d len0 %d\n", beam, rc, len[0]);
  }
  {   // probing-layout model (trigram sample)
    std::string plm = std::string(G) + "toy_data_lm.binary";
    void* ps = nullptr;
    rc = os2s_ctc_dict_scorer_create(plm.c_str(), vp.data(), (int)vp.size(), 1.0, 0.3, &ps);
    int cb = -1, mo = -1, dsz = -1;
    os2s_ctc_dict_scorer_info(ps, &cb, &mo, &dsz);
    printf("probing dict scorer rc %d order %d dict %d\n", rc, mo, dsz);
    std::vector<int32_t> ids((size_t)B * T), len(B); std::vector<float> lp(B);
    rc = os2s_ctc_dict_beam_search(probs.data(), B * C, C, sl.data(), T, B, C, 16, 1.0, 40, 1, ps, 2, ids.data(), len.data(), lp.data());
    printf("probing dict decode rc %d\n", rc);
    os2s_ctc_dict_scorer_destroy(ps);
  }
  const char* words[2] = {"ten", "seconds"}; float p = 0;
  os2s_ctc_scorer_ngram_score(sc, words, 2, &p); printf("ngram %f\n", p);
  rc = os2s_ctc_generate_trie("/tmp/alpha.txt", lm.c_str(), "/tmp/alpha.txt", "/tmp/out.trie");
  printf("generate_trie rc %d\n", rc);
  os2s_ctc_scorer_destroy(sc); os2s_ctc_dict_scorer_destroy(ds);
  return 0;
}
