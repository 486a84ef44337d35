// probe: lane-0 successive-shortest-path EMD in LDS vs the same code on the host
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
#include <vector>
#define NN 16
#define MM 64
template <typename FLAG>
__host__ __device__ double ssp(int n, int m, const double* a, const double* b, const double* C,
    double* fl, double* sup, double* dem, double* pot, double* dist, int* prevn, FLAG* done) {
  const int N = n + m; const double EPS = 1e-13;
  for (int i = 0; i < n; i++) sup[i] = a[i];
  for (int j = 0; j < m; j++) dem[j] = b[j];
  for (int i = 0; i < n * m; i++) fl[i] = 0.0;
  for (int i = 0; i < N; i++) pot[i] = 0.0;
  for (int iter = 0; iter < 100000; iter++) {
    int any = 0;
    for (int i = 0; i < N; i++) { dist[i] = INFINITY; prevn[i] = -1; done[i] = 0; }
    for (int i = 0; i < n; i++) if (sup[i] > EPS) { dist[i] = 0.0; any = 1; }
    if (!any) break;
    int any_dem = 0;
    for (int j = 0; j < m; j++) if (dem[j] > EPS) any_dem = 1;
    if (!any_dem) break;
    int target = -1;
    bool searching = true;
    while (searching) {
      int x = -1; double bd = INFINITY;
      for (int i = 0; i < N; i++) if (!done[i] && dist[i] < bd) { bd = dist[i]; x = i; }
      if (x < 0) { searching = false; }
      else {
        done[x] = 1;
        if (x >= n && dem[x - n] > EPS) { target = x; searching = false; }
        else if (x < n) {
          for (int j = 0; j < m; j++) {
            if (done[n + j]) continue;
            double rc = C[x * m + j] + pot[x] - pot[n + j]; if (rc < 0) rc = 0;
            if (dist[x] + rc < dist[n + j]) { dist[n + j] = dist[x] + rc; prevn[n + j] = x; }
          }
        } else {
          const int j = x - n;
          for (int i = 0; i < n; i++) {
            if (done[i] || !(fl[i * m + j] > EPS)) continue;
            double rc = -C[i * m + j] + pot[x] - pot[i]; if (rc < 0) rc = 0;
            if (dist[x] + rc < dist[i]) { dist[i] = dist[x] + rc; prevn[i] = x; }
          }
        }
      }
    }
    if (target < 0) break;
    const double dt = dist[target];
    for (int i = 0; i < N; i++) pot[i] += (done[i] && dist[i] < dt) ? dist[i] : dt;
    double delta = dem[target - n];
    int x = target;
    while (prevn[x] >= 0) { const int pr = prevn[x]; if (pr >= n) { const double cap = fl[x * m + (pr - n)]; if (cap < delta) delta = cap; } x = pr; }
    if (sup[x] < delta) delta = sup[x];
    sup[x] -= delta; dem[target - n] -= delta;
    x = target;
    while (prevn[x] >= 0) { const int pr = prevn[x]; if (pr < n) fl[pr * m + (x - n)] += delta; else fl[x * m + (pr - n)] -= delta; x = pr; }
  }
  double cost = 0; for (int i = 0; i < n * m; i++) cost += fl[i] * C[i];
  return cost;
}
template <typename FLAG>
__global__ void k(int n, int m, const double* a, const double* b, const double* C, double* out) {
  __shared__ double fl[NN * MM], sup[NN], dem[MM], pot[NN + MM], dist[NN + MM];
  __shared__ int prevn[NN + MM];
  __shared__ FLAG done[NN + MM];
  __syncthreads();
  if (threadIdx.x != 0) return;
  out[blockIdx.x] = ssp<FLAG>(n, m, a, b, C + blockIdx.x * n * m, fl, sup, dem, pot, dist, prevn, done);
}
int main() {
  const int n = 3, m = 7, P = 8;
  std::vector<double> a(n), b(m), C(P * n * m), out(P), out2(P);
  srand(5);
  double sa = 0, sb = 0;
  for (auto& x : a) { x = 0.1 + rand() / (double)RAND_MAX; sa += x; } for (auto& x : a) x /= sa;
  for (auto& x : b) { x = 0.1 + rand() / (double)RAND_MAX; sb += x; } for (auto& x : b) x /= sb;
  for (auto& x : C) x = rand() / (double)RAND_MAX;
  double *da, *db, *dC, *dout;
  hipMalloc(&da, n * 8); hipMalloc(&db, m * 8); hipMalloc(&dC, C.size() * 8); hipMalloc(&dout, P * 8);
  hipMemcpy(da, a.data(), n * 8, hipMemcpyHostToDevice); hipMemcpy(db, b.data(), m * 8, hipMemcpyHostToDevice);
  hipMemcpy(dC, C.data(), C.size() * 8, hipMemcpyHostToDevice);
  k<unsigned char><<<P, 64>>>(n, m, da, db, dC, dout);
  hipMemcpy(out.data(), dout, P * 8, hipMemcpyDeviceToHost);
  k<int><<<P, 64>>>(n, m, da, db, dC, dout);
  hipMemcpy(out2.data(), dout, P * 8, hipMemcpyDeviceToHost);
  std::vector<double> fl(NN * MM), sup(NN), dem(MM), pot(NN + MM), dist(NN + MM); std::vector<int> prevn(NN + MM), done(NN + MM);
  for (int p = 0; p < P; p++) {
    double ref = ssp<int>(n, m, a.data(), b.data(), C.data() + p * n * m, fl.data(), sup.data(), dem.data(), pot.data(), dist.data(), prevn.data(), done.data());
    printf("problem %d host %.12g gpu(u8) %.12g gpu(int) %.12g\n", p, ref, out[p], out2[p]);
  }
  return 0;
}
