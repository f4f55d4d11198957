// trace_step.inl -- ONE traversal step of a wavefront, included textually in the loop of both trace kernels (k_trace and
// k_trace_level in kernels.hip), which declare the state it works on: cur, sp, fl, oo, od, oi, lim_n, slack_abs, o_key,
// o_slot, o_tri, e_ray (per lane), tid, lane, s_stack, push / pop, g_nodes, g_qnodes, g_tris, delta, QN (which node format),
// the c_* counters and the F_* flag bits.
        // ---- one step per iteration, chosen for the WHOLE wavefront: either every lane that stands on an inner node
        // visits it, or every lane that stands on a leaf tests ONE of its triangles.  The kind with more waiting lanes
        // runs (node steps weighted by NODE_BIAS/8); the other lanes keep their state for a later iteration.  A step
        // costs the same whether 1 or 64 lanes need it, so this keeps the SIMD lanes of divergent rays busy.
        const unsigned long long m_inner = __ballot(cur < REF_LEAF);
        const unsigned long long m_leaf = __ballot(cur >= REF_LEAF && cur < REF_IDLE);
        const bool node_step = m_inner != 0ull && (__popcll(m_inner) * NODE_BIAS >= __popcll(m_leaf) * 8);

        if (node_step) {
            if (cur < REF_LEAF) {
                if (COUNT) {
                    c_nodes++; c_boxes += 4;
                    const unsigned long long am = __ballot(true);
                    const uint32_t first_lane = (uint32_t)__ffsll((long long)am) - 1u;
                    if (lane == first_lane) c_wnode++;
                    // node steps in which every participating lane stands on the SAME node
                    const uint32_t cur0 = (uint32_t)__builtin_amdgcn_readfirstlane((int)cur);
#if defined(TRR_COUNT_DISTINCT)
                    {   // experiment: distinct nodes among the lanes of this node step (what a shared-front traversal would visit one by one)
                        unsigned long long left = am;
                        uint32_t distinct = 0;
                        while (left != 0ull) {
                            const uint32_t l0 = (uint32_t)__ffsll((long long)left) - 1u;
                            const uint32_t c0 = (uint32_t)__shfl((int)cur, (int)l0);
                            left &= ~__ballot(cur == c0);
                            ++distinct;
                        }
                        if (lane == first_lane) c_wuni += distinct;
                    }
#elif !defined(TRR_COUNT_WAIT)
                    if (__ballot(cur == cur0) == am && lane == first_lane) c_wuni++;
#endif
                }
                float t0, t1, t2, t3;
                uint32_t r0, r1, r2, r3;
#ifdef TRR_WIDE8
                if constexpr (QN) {
                    // experiment: 8-wide exact node (bvh.h), fourteen 16-B loads, eight slab tests, 19-comparator sort, up to seven pushes
                    if (COUNT) c_boxes += 4;
                    const float4* nd = reinterpret_cast<const float4*>(g_qnodes + cur);
                    const float4 lxa = nd[0], lxb = nd[1], lya = nd[2], lyb = nd[3], lza = nd[4], lzb = nd[5];
                    const float4 hxa = nd[6], hxb = nd[7], hya = nd[8], hyb = nd[9], hza = nd[10], hzb = nd[11];
                    const uint4 cha = *reinterpret_cast<const uint4*>(nd + 12), chb = *reinterpret_cast<const uint4*>(nd + 13);
                    float t4, t5, t6, t7;
                    uint32_t r4, r5, r6, r7;
#define TRR_BOX(T, R, LX, LY, LZ, HX, HY, HZ, CH)                                                          \
    {                                                                                                      \
        float ax = ((LX) - oo.x) * oi.x, bx = ((HX) - oo.x) * oi.x;                                        \
        float ay = ((LY) - oo.y) * oi.y, by = ((HY) - oo.y) * oi.y;                                        \
        float az = ((LZ) - oo.z) * oi.z, bz = ((HZ) - oo.z) * oi.z;                                        \
        float tmin = rmax(rmax(rmin(ax, bx), rmin(ay, by)), rmin(az, bz));                                 \
        float tmax = rmin(rmin(rmax(ax, bx), rmax(ay, by)), rmax(az, bz));                                 \
        bool hit = (tmax > rmax(tmin, 0.0f)) & !(tmin > lim_n);                                            \
        T = hit ? tmin : INFINITY;                                                                         \
        R = hit ? (CH) : REF_EMPTY;                                                                        \
    }
                    TRR_BOX(t0, r0, lxa.x, lya.x, lza.x, hxa.x, hya.x, hza.x, cha.x)
                    TRR_BOX(t1, r1, lxa.y, lya.y, lza.y, hxa.y, hya.y, hza.y, cha.y)
                    TRR_BOX(t2, r2, lxa.z, lya.z, lza.z, hxa.z, hya.z, hza.z, cha.z)
                    TRR_BOX(t3, r3, lxa.w, lya.w, lza.w, hxa.w, hya.w, hza.w, cha.w)
                    TRR_BOX(t4, r4, lxb.x, lyb.x, lzb.x, hxb.x, hyb.x, hzb.x, chb.x)
                    TRR_BOX(t5, r5, lxb.y, lyb.y, lzb.y, hxb.y, hyb.y, hzb.y, chb.y)
                    TRR_BOX(t6, r6, lxb.z, lyb.z, lzb.z, hxb.z, hyb.z, hzb.z, chb.z)
                    TRR_BOX(t7, r7, lxb.w, lyb.w, lzb.w, hxb.w, hyb.w, hzb.w, chb.w)
#undef TRR_BOX
#define TRR_CSWAP(TI, RI, TJ, RJ)                                                  \
    {                                                                              \
        bool sw_ = TJ < TI;                                                        \
        float ta_ = sw_ ? TJ : TI, tb_ = sw_ ? TI : TJ;                            \
        uint32_t ra_ = sw_ ? RJ : RI, rb_ = sw_ ? RI : RJ;                         \
        TI = ta_; TJ = tb_; RI = ra_; RJ = rb_;                                    \
    }
                    // a NaN tmin (kept child) compares false both ways and may end up anywhere; the pushes below therefore test
                    // every entry instead of relying on "hits first"
                    TRR_CSWAP(t0, r0, t1, r1) TRR_CSWAP(t2, r2, t3, r3) TRR_CSWAP(t4, r4, t5, r5) TRR_CSWAP(t6, r6, t7, r7)
                    TRR_CSWAP(t0, r0, t2, r2) TRR_CSWAP(t1, r1, t3, r3) TRR_CSWAP(t4, r4, t6, r6) TRR_CSWAP(t5, r5, t7, r7)
                    TRR_CSWAP(t1, r1, t2, r2) TRR_CSWAP(t5, r5, t6, r6)
                    TRR_CSWAP(t0, r0, t4, r4) TRR_CSWAP(t1, r1, t5, r5) TRR_CSWAP(t2, r2, t6, r6) TRR_CSWAP(t3, r3, t7, r7)
                    TRR_CSWAP(t2, r2, t4, r4) TRR_CSWAP(t3, r3, t5, r5)
                    TRR_CSWAP(t1, r1, t2, r2) TRR_CSWAP(t3, r3, t4, r4) TRR_CSWAP(t5, r5, t6, r6)
#undef TRR_CSWAP
                    if (sp + 7 <= STACK_LDS) {
                        int at = sp;
                        s_stack[at * TRACE_THREADS + tid] = make_uint2(r7, __float_as_uint(t7)); at += (r7 != REF_EMPTY ? 1 : 0);
                        s_stack[at * TRACE_THREADS + tid] = make_uint2(r6, __float_as_uint(t6)); at += (r6 != REF_EMPTY ? 1 : 0);
                        s_stack[at * TRACE_THREADS + tid] = make_uint2(r5, __float_as_uint(t5)); at += (r5 != REF_EMPTY ? 1 : 0);
                        s_stack[at * TRACE_THREADS + tid] = make_uint2(r4, __float_as_uint(t4)); at += (r4 != REF_EMPTY ? 1 : 0);
                        s_stack[at * TRACE_THREADS + tid] = make_uint2(r3, __float_as_uint(t3)); at += (r3 != REF_EMPTY ? 1 : 0);
                        s_stack[at * TRACE_THREADS + tid] = make_uint2(r2, __float_as_uint(t2)); at += (r2 != REF_EMPTY ? 1 : 0);
                        s_stack[at * TRACE_THREADS + tid] = make_uint2(r1, __float_as_uint(t1)); at += (r1 != REF_EMPTY ? 1 : 0);
                        sp = at;
                    } else {
                        if (r7 != REF_EMPTY) push(r7, t7);
                        if (r6 != REF_EMPTY) push(r6, t6);
                        if (r5 != REF_EMPTY) push(r5, t5);
                        if (r4 != REF_EMPTY) push(r4, t4);
                        if (r3 != REF_EMPTY) push(r3, t3);
                        if (r2 != REF_EMPTY) push(r2, t2);
                        if (r1 != REF_EMPTY) push(r1, t1);
                    }
                    cur = (r0 != REF_EMPTY) ? r0 : pop();
                } else
#endif
                if constexpr (QN) {
                    // ---- 64-B node (bvh.h QNode4): four 16-B loads per lane instead of seven.  Child k spans
                    // [org + qlo_k s, org + qhi_k s] per axis, a superset of its exact box; the slab distances come out of
                    // ONE fused multiply-add per plane, t = q * (s / d) + (org -+ e - o) / d.  That is not the arithmetic of
                    // the triangle's own-box test below (which stays exact), so every box is pushed out by e = e_ray first:
                    // 2^-20 of the largest coordinate in play, several times the rounding error of either form, hence a
                    // near distance never above and a far distance never below what the exact test of anything inside the
                    // box can compute -- the node test stays a superset test (candidate-set invariant).  1/d is
                    // clamped to +-2^60 for this test only: beyond that a distance is "never" either way, and the
                    // products cannot overflow into inf - inf.
                    const float4* nd = reinterpret_cast<const float4*>(g_qnodes + cur);
                    const float4 c0 = nd[0], c1 = nd[1], c2 = nd[2];
                    const uint4 ch = *reinterpret_cast<const uint4*>(nd + 3);
                    const float ix = __builtin_amdgcn_fmed3f(oi.x, -0x1p60f, 0x1p60f), iy = __builtin_amdgcn_fmed3f(oi.y, -0x1p60f, 0x1p60f),
                                iz = __builtin_amdgcn_fmed3f(oi.z, -0x1p60f, 0x1p60f);
                    const float Ax = c0.w * ix, Ay = c1.x * iy, Az = c1.y * iz;
                    const float dx = c0.x - oo.x, dy = c0.y - oo.y, dz = c0.z - oo.z;
                    const float Blx = (dx - e_ray) * ix, Bhx = (dx + e_ray) * ix;
                    const float Bly = (dy - e_ray) * iy, Bhy = (dy + e_ray) * iy;
                    const float Blz = (dz - e_ray) * iz, Bhz = (dz + e_ray) * iz;
                    const uint32_t qlx = __float_as_uint(c1.z), qly = __float_as_uint(c1.w), qlz = __float_as_uint(c2.x);
                    const uint32_t qhx = __float_as_uint(c2.y), qhy = __float_as_uint(c2.z), qhz = __float_as_uint(c2.w);
#define TRR_QBOX(T, R, K, CH)                                                                                     \
    {                                                                                                             \
        float ax = __builtin_fmaf((float)((qlx >> (8 * K)) & 0xFFu), Ax, Blx), bx = __builtin_fmaf((float)((qhx >> (8 * K)) & 0xFFu), Ax, Bhx); \
        float ay = __builtin_fmaf((float)((qly >> (8 * K)) & 0xFFu), Ay, Bly), by = __builtin_fmaf((float)((qhy >> (8 * K)) & 0xFFu), Ay, Bhy); \
        float az = __builtin_fmaf((float)((qlz >> (8 * K)) & 0xFFu), Az, Blz), bz = __builtin_fmaf((float)((qhz >> (8 * K)) & 0xFFu), Az, Bhz); \
        float tmin = rmax(rmax(rmin(ax, bx), rmin(ay, by)), rmin(az, bz));                                        \
        float tmax = rmin(rmin(rmax(ax, bx), rmax(ay, by)), rmax(az, bz));                                        \
        bool hit = (tmax > rmax(tmin, 0.0f)) & !(tmin > lim_n);                                                   \
        T = hit ? tmin : INFINITY;                                                                                \
        R = hit ? (CH) : REF_EMPTY;                                                                               \
    }
                    TRR_QBOX(t0, r0, 0, ch.x)
                    TRR_QBOX(t1, r1, 1, ch.y)
                    TRR_QBOX(t2, r2, 2, ch.z)
                    TRR_QBOX(t3, r3, 3, ch.w)
#undef TRR_QBOX
                } else {
#ifdef TRR_UNIFORM_SCALAR
                    // experiment (kept out): when every lane of the step stands on the SAME node, fetch it once
                    // through the scalar cache instead of 64 x 112 B through the vector memory path
                    float4 lox, loy, loz, hix, hiy, hiz;
                    uint4 ch;
                    {
                        const uint32_t cur_first = (uint32_t)__builtin_amdgcn_readfirstlane((int)cur);
                        if (__ballot(cur == cur_first) == __ballot(true)) {
                            typedef const float __attribute__((address_space(4))) * cfp;
                            typedef const uint32_t __attribute__((address_space(4))) * cup;
                            cfp sf = (cfp)(uintptr_t)(g_nodes + cur_first);
                            cup su = (cup)(uintptr_t)(g_nodes + cur_first);
                            lox = make_float4(sf[0], sf[1], sf[2], sf[3]); loy = make_float4(sf[4], sf[5], sf[6], sf[7]);
                            loz = make_float4(sf[8], sf[9], sf[10], sf[11]); hix = make_float4(sf[12], sf[13], sf[14], sf[15]);
                            hiy = make_float4(sf[16], sf[17], sf[18], sf[19]); hiz = make_float4(sf[20], sf[21], sf[22], sf[23]);
                            ch = make_uint4(su[24], su[25], su[26], su[27]);
                        } else {
                            const float4* nd = reinterpret_cast<const float4*>(g_nodes + cur);
                            lox = nd[0]; loy = nd[1]; loz = nd[2]; hix = nd[3]; hiy = nd[4]; hiz = nd[5];
                            ch = *reinterpret_cast<const uint4*>(nd + 6);
                        }
                    }
#else
                    const float4* nd = reinterpret_cast<const float4*>(g_nodes + cur);
                    float4 lox = nd[0], loy = nd[1], loz = nd[2], hix = nd[3], hiy = nd[4], hiz = nd[5];
                    uint4 ch = *reinterpret_cast<const uint4*>(nd + 6);
#endif
#ifdef TRR_DUMMY_LOAD
                    {   // experiment: TRR_DUMMY_LOAD more 16-B loads per lane from the node's own cache line, through a pointer the
                        // optimiser cannot see through (is the vector memory path a limit, and how steep is it?)
                        const float4* xp = reinterpret_cast<const float4*>(g_nodes + cur);
                        asm volatile("" : "+v"(xp));
#pragma unroll
                        for (int k = 0; k < TRR_DUMMY_LOAD; ++k) {
                            float4 extra = xp[(7 + k) & 7];
                            asm volatile("" ::"v"(extra.x), "v"(extra.y), "v"(extra.z), "v"(extra.w));
                        }
                    }
#endif
#ifdef TRR_DUMMY_VALU
                    {   // experiment: TRR_DUMMY_VALU extra dependent vector instructions per node step (is the kernel VALU-bound?)
                        float dv = lox.x;
#pragma unroll
                        for (int k = 0; k < TRR_DUMMY_VALU; ++k) asm volatile("v_add_f32 %0, %0, %0" : "+v"(dv));
                        asm volatile("" ::"v"(dv));
                    }
#endif
                    // [crate] Ray::intersects_aabb per child, same arithmetic as the per-triangle test below, so a parent
                    // box (exact union of its children) can never reject what a triangle's own box accepts.  Written with
                    // scalars and selects only (no arrays, no short-circuit branches): keeps the step branch-free.
                    // An unused child slot holds the box (+inf .. +inf), which fails the test for every ray; a NaN tmin keeps
                    // the child (NaN > x is false) and simply does not take part in the ordering.
#define TRR_BOX(T, R, LX, LY, LZ, HX, HY, HZ, CH)                                                          \
    {                                                                                                      \
        float ax = ((LX) - oo.x) * oi.x, bx = ((HX) - oo.x) * oi.x;                                        \
        float ay = ((LY) - oo.y) * oi.y, by = ((HY) - oo.y) * oi.y;                                        \
        float az = ((LZ) - oo.z) * oi.z, bz = ((HZ) - oo.z) * oi.z;                                        \
        float tmin = rmax(rmax(rmin(ax, bx), rmin(ay, by)), rmin(az, bz));                                 \
        float tmax = rmin(rmin(rmax(ax, bx), rmax(ay, by)), rmax(az, bz));                                 \
        bool hit = (tmax > rmax(tmin, 0.0f)) & !(tmin > lim_n);                                            \
        T = hit ? tmin : INFINITY;                                                                         \
        R = hit ? (CH) : REF_EMPTY;                                                                        \
    }
                    TRR_BOX(t0, r0, lox.x, loy.x, loz.x, hix.x, hiy.x, hiz.x, ch.x)
                    TRR_BOX(t1, r1, lox.y, loy.y, loz.y, hix.y, hiy.y, hiz.y, ch.y)
                    TRR_BOX(t2, r2, lox.z, loy.z, loz.z, hix.z, hiy.z, hiz.z, ch.z)
                    TRR_BOX(t3, r3, lox.w, loy.w, loz.w, hix.w, hiy.w, hiz.w, ch.w)
#undef TRR_BOX
                }
#ifdef TRR_WIDE8
                if constexpr (!QN) {
#else
                {
#endif
                // sort the four (tmin, ref) pairs ascending: 5-comparator network of selects
#define TRR_CSWAP(TI, RI, TJ, RJ)                                                  \
    {                                                                              \
        bool sw_ = TJ < TI;                                                        \
        float ta_ = sw_ ? TJ : TI, tb_ = sw_ ? TI : TJ;                            \
        uint32_t ra_ = sw_ ? RJ : RI, rb_ = sw_ ? RI : RJ;                         \
        TI = ta_; TJ = tb_; RI = ra_; RJ = rb_;                                    \
    }
                TRR_CSWAP(t0, r0, t1, r1) TRR_CSWAP(t2, r2, t3, r3) TRR_CSWAP(t0, r0, t2, r2)
                TRR_CSWAP(t1, r1, t3, r3) TRR_CSWAP(t1, r1, t2, r2)
#undef TRR_CSWAP
                // far children go on the stack, farthest first.  Fast path: three unconditional LDS stores, the stack
                // pointer advances only past valid entries (a slot above the top may hold garbage).
                if (sp + 3 <= STACK_LDS) {
                    const int i3 = sp, i2 = i3 + (r3 != REF_EMPTY ? 1 : 0), i1 = i2 + (r2 != REF_EMPTY ? 1 : 0);
                    s_stack[i3 * TRACE_THREADS + tid] = make_uint2(r3, __float_as_uint(t3));
                    s_stack[i2 * TRACE_THREADS + tid] = make_uint2(r2, __float_as_uint(t2));
                    s_stack[i1 * TRACE_THREADS + tid] = make_uint2(r1, __float_as_uint(t1));
                    sp = i1 + (r1 != REF_EMPTY ? 1 : 0);
                } else {
                    if (r3 != REF_EMPTY) push(r3, t3);
                    if (r2 != REF_EMPTY) push(r2, t2);
                    if (r1 != REF_EMPTY) push(r1, t1);
                }
                cur = (r0 != REF_EMPTY) ? r0 : pop();
                }
            }
        } else if (m_leaf != 0ull) {
            // ---- leaf: own-box test + exact test of the leaf's next TRR_TRIS_PER_STEP remaining triangles ------------
            if (cur >= REF_LEAF && cur < REF_IDLE) {
                const uint32_t slot = (cur & 0x7FFFFFFFu) >> 2;
                if (COUNT) {
                    if (lane == (uint32_t)__ffsll((long long)__ballot(true)) - 1u) c_wtri++;
                }
                // one triangle: triangle.rs:88-95 box (empty.grow(a).grow(b).grow(c + 1e-4)) + [crate] slab test, then the
                // exact test; candidates are ranked by (key, triangle id), so the order of the tests does not matter
                auto test_one = [&](uint32_t sl, float4 q0, float4 q1, float4 q2, float4 q3) {
                    if (COUNT) c_boxes++;
                    bool cand = true;
                    if (!(fl & F_SKIPBOX)) {
                        F3 a = f3(q0.x, q0.y, q0.z), b = f3(q0.w, q1.x, q1.y), c = f3(q1.z, q1.w, q2.x);
                        F3 c2 = f3(c.x + 0.0001f, c.y + 0.0001f, c.z + 0.0001f);
                        F3 lo = f3(rmin(rmin(a.x, b.x), c2.x), rmin(rmin(a.y, b.y), c2.y), rmin(rmin(a.z, b.z), c2.z));
                        F3 hi = f3(rmax(rmax(a.x, b.x), c2.x), rmax(rmax(a.y, b.y), c2.y), rmax(rmax(a.z, b.z), c2.z));
                        float ax = (lo.x - oo.x) * oi.x, bx = (hi.x - oo.x) * oi.x;
                        float ay = (lo.y - oo.y) * oi.y, by = (hi.y - oo.y) * oi.y;
                        float az = (lo.z - oo.z) * oi.z, bz = (hi.z - oo.z) * oi.z;
                        float tmin = rmax(rmax(rmin(ax, bx), rmin(ay, by)), rmin(az, bz));
                        float tmax = rmin(rmin(rmax(ax, bx), rmax(ay, by)), rmax(az, bz));
                        // the second condition: cannot beat the current best (slack keeps this conservative)
                        cand = (tmax > rmax(tmin, 0.0f)) && !(tmin > lim_n);
                    }
                    if (cand) {
                        if (COUNT) c_tris++;
                        TriHit h;
                        if (tri_test(q0, q1, q2, q3.z, oo, od, delta, h)) {
                            uint32_t tri_id = __float_as_uint(q3.x);
                            if (candidate_better((fl & F_OHAVE) != 0u, h.key, tri_id, o_key, o_tri)) {
                                fl |= F_OHAVE; o_key = h.key; o_slot = sl; o_tri = tri_id;
                                float dist = sqrtf(h.key);
                                if (dist == dist) lim_n = rmin(lim_n, dist * (1.0f + PRUNE_REL) + slack_abs);
                            }
                        }
                    }
                };
                const float4* tp = reinterpret_cast<const float4*>(g_tris + slot);
#if TRR_TRIS_PER_STEP >= 2
                // both records are requested before the first is used: the second one's latency hides behind the first test
                uint32_t sl = slot;
                const bool two = (cur & 3u) != 0u;
                const float4* tq = two ? tp + 4 : tp;
                float4 a0 = tp[0], a1 = tp[1], a2 = tp[2], a3 = tp[3];
                float4 b0 = tq[0], b1 = tq[1], b2 = tq[2], b3 = tq[3];
                test_one(sl, a0, a1, a2, a3);
                if (two) test_one(sl + 1u, b0, b1, b2, b3);
                // ref = LEAF | first << 2 | count-1: two triangles consumed -> first+2, count-2
                cur = ((cur & 3u) >= 2u) ? cur + 6u : pop();
#else
                test_one(slot, tp[0], tp[1], tp[2], tp[3]);
                // next triangle of the leaf (ref = LEAF | first << 2 | count-1  ->  first+1, count-2), else pop
                cur = (cur & 3u) ? cur + 3u : pop();
#endif
            }
        }
