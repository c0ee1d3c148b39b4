// Sanitizer driver for the internal tree's builder (pt_tree_sweep.h is host code): random clouds, degenerate inputs and a
// size that takes the threaded path; every tree is checked to be a cover of the primitives with boxes that are unions.
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <random>
#include <vector>

#include "pt_tree_sweep.h"

static int check(const std::vector<float>& boxes, int n) {
    std::vector<pt_bvh_node> out;
    int32_t root = -1, depth = 0;
    pts::build_sweep_tree(boxes.data(), n, out, &root, &depth);
    if ((int)out.size() != 2 * n - 1 || root != 0) return 1;
    std::vector<char> seen(n, 0);
    std::vector<int> todo(1, root);
    size_t visited = 0;
    while (!todo.empty()) {
        const int k = todo.back();
        todo.pop_back();
        if (++visited > out.size()) return 2;
        const pt_bvh_node& nd = out[k];
        if (nd.prim >= 0) {
            if (nd.prim >= n || seen[nd.prim] || nd.left != -1 || nd.right != -1) return 3;
            seen[nd.prim] = 1;
            for (int a = 0; a < 3; a++)
                if (nd.bmin[a] != boxes[(size_t)nd.prim * 6 + a] || nd.bmax[a] != boxes[(size_t)nd.prim * 6 + 3 + a]) return 4;
            continue;
        }
        if (nd.left < 0 || nd.right < 0 || nd.left >= (int)out.size() || nd.right >= (int)out.size()) return 5;
        for (int a = 0; a < 3; a++) {
            if (nd.bmin[a] != std::fmin(out[nd.left].bmin[a], out[nd.right].bmin[a])) return 6;
            if (nd.bmax[a] != std::fmax(out[nd.left].bmax[a], out[nd.right].bmax[a])) return 7;
        }
        todo.push_back(nd.left);
        todo.push_back(nd.right);
    }
    if (visited != out.size()) return 8;
    for (int i = 0; i < n; i++)
        if (!seen[i]) return 9;
    return 0;
}

int main() {
    std::mt19937 g(7);
    std::uniform_real_distribution<float> U(0.0f, 1.0f);
    for (int n : {1, 2, 3, 7, 64, 1000, 70000}) {
        for (int kind = 0; kind < 4; kind++) {
            std::vector<float> b((size_t)n * 6);
            for (int i = 0; i < n; i++)
                for (int k = 0; k < 3; k++) {
                    float c = kind == 1 ? 0.5f : U(g) * 10.0f;                        // kind 1: every box the same
                    float e = kind == 2 ? std::ldexp(1.0f, i % 60) : kind == 3 ? 0.0f : 0.05f * U(g);   // 2: nested shells, 3: points
                    b[(size_t)i * 6 + k] = kind == 2 ? -e : c;
                    b[(size_t)i * 6 + 3 + k] = kind == 2 ? e : c + e;
                }
            const int rc = check(b, n);
            if (rc) { std::printf("n %d kind %d: check failed with %d\n", n, kind, rc); return 1; }
        }
    }
    std::printf("sweep sanitizer driver done\n");
    return 0;
}
