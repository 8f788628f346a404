// Host step drivers (included by engine.hip).  Flag logic follows the reference loops line by line:
//   SD  : models/region_diffusion.py:99-173        XL : models/region_diffusion_sdxl.py:779-872

static void pndm_coeffs(rt_engine* e, int i, StepArgs& a) {
    // PNDMScheduler.step_plms (diffusers 0.18.2, [memory]); see oracle/schedulers.py:OraclePNDM
    const int ratio = 1000 / e->num_inference_steps;
    int t = (int)e->timesteps[i];
    int prev_t = t - ratio;
    a.push = 0;
    if (e->pndm_counter != 1) {
        a.push = 1;
        if (e->pndm_nets < 4) e->pndm_nets++;
        e->pndm_head = (e->pndm_head + 3) & 3;        // new newest slot
    } else {
        prev_t = t; t = t + ratio;
    }
    const size_t per = (size_t)2 * 4 * e->lat_h * e->lat_w;
    for (int k = 0; k < 4; ++k) a.ets[k] = e->ets + (size_t)((e->pndm_head + k) & 3) * per;
    if (!a.push) { for (int k = 3; k >= 1; --k) a.ets[k] = a.ets[k - 1]; }   // ets[1] = newest stored
    if (e->pndm_nets == 1 && e->pndm_counter == 0) a.pndm_mode = 0;
    else if (e->pndm_nets == 1 && e->pndm_counter == 1) a.pndm_mode = 1;
    else a.pndm_mode = e->pndm_nets;   // 2, 3, 4
    RT_REQUIRE((int)e->table.size() >= 1000, "pndm: alphas_cumprod table missing");
    const double a_t = e->table[t], a_p = prev_t >= 0 ? e->table[prev_t] : e->table[0];
    const double b_t = 1 - a_t, b_p = 1 - a_p;
    // computed in fp32 like the reference's tensor arithmetic on fp32 alphas
    const float fa_t = (float)a_t, fa_p = (float)a_p, fb_t = (float)b_t, fb_p = (float)b_p;
    const float sample_coeff = std::sqrt(fa_p / fa_t);
    const float denom = fa_t * std::sqrt(fb_p) + std::sqrt(fa_t * fb_t * fa_p);
    a.ca = sample_coeff; a.cb = (fa_p - fa_t) / denom;
    a.cur_sample = e->cur_sample;
    e->pndm_counter++;
}

// The streams of rich-text step i and the flags of its epilogue (everything of region_step that is a pure function of the schedule
// position): `in` = the F batched forwards with their mode words, `a` = the epilogue's arguments without the scheduler coefficients
// (PNDM's are stateful: region_finish computes them once).
void rt_engine::region_plan(int i, float g, double isa, double ibg, bool xl, bool elide, bool defer_blend, FwdIn& in, StepArgs& a, bool& blend_out,
                            bool& inject_out) {
    require_bound();
    const int n = (int)timesteps.size(), R = n_regions;
    RT_REQUIRE(i >= 0 && i < n, "region_step: step index out of range");
    RT_REQUIRE(R >= 1 && n_prompts == R + 1, "region_step: need R masks and R+1 prompts (rd.py:96-97)");
    RT_REQUIRE(mask_hw == lat_h * lat_w && lat_h > 0, "region_step: masks/latents shape mismatch");
    RT_REQUIRE((sched_kind == RT_SCHED_EULER) == xl, "region_step: SD uses PNDM, SDXL uses Euler");
    const float t = timesteps[i];
    const bool use_ref = isa > 0 || ibg > 0;
    // `t > (1-inject_selfattn)*1000`: torch compares a float32 / int64 tensor element with a Python float in float32
    const float thr = (float)((1.0 - isa) * 1000.0);
    auto feat_at = [&](int j) { return timesteps[j] > thr; };
    const bool feat = feat_at(i);
    const int bg_index = (int)(ibg * (double)n);                   // int(inject_background * len(timesteps)) on Python floats
    const bool blend = (i == bg_index) && ibg > 0;
    bool step_ref = use_ref;
    if (xl) step_ref = isa > 0 || ((double)i < ibg * (double)n);
    bool run_ref = use_ref;
    if (use_ref && elide) {
        // the reference pair can only influence the output through injection at this step, or through
        // latents_reference consumed by a later injected step or by the blend
        int last_use = ibg > 0 ? bg_index : -1;
        for (int j = 0; j < n; ++j) if (feat_at(j)) last_use = std::max(last_use, j);
        run_ref = i <= last_use;
    }
    if (!run_ref) step_ref = false;

    in = FwdIn{}; in.h = lat_h; in.w = lat_w; in.t = t; in.eps_out = eps;
    const float scale = xl ? 1.f / std::sqrt(table[i] * table[i] + 1.f) : 1.f;
    a = StepArgs{};
    int F = 0;
    auto add = [&](const float* x, int prompt, int fs) {
        RT_REQUIRE(F < cfg.max_streams, "region_step: more streams than max_streams");
        in.x[F] = x; in.scale[F] = scale; in.prompt[F] = prompt; in.fontsize[F] = fs; in.qk_src[F] = F; in.res_src[F] = -1;
        return F++;
    };
    a.s_uncond = add(lat, 0, 0);
    a.s_base = add(lat, R, 1);
    a.s_uref = a.s_tref = -1;
    if (run_ref) { a.s_uref = add(lat_ref, 0, 0); a.s_tref = add(lat_ref, R, 0); }
    for (int r = 0; r < R - 1; ++r) {
        const int s = add(lat, r + 1, 0);
        if (feat && run_ref) { in.qk_src[s] = a.s_tref; in.res_src[s] = a.s_tref; }
        a.s_region[r] = s;
    }
    in.B = F;
    a.eps = eps; a.masks = masks; a.lat = lat; a.lat_ref = lat_ref; a.HW = lat_h * lat_w; a.R = R; a.g = g; a.plain = 0;
    a.noise_pred = noise_pred;
    a.sched = sched_kind; a.step_ref = step_ref ? 1 : 0; a.blend = (blend && !defer_blend) ? 1 : 0;
    blend_out = blend && defer_blend;
    inject_out = feat && run_ref;
}

// mask combine + CFG + scheduler step + blend on the eps of ALL streams of step i (models/region_diffusion.py:119-147,
// region_diffusion_sdxl.py:810-846)
void rt_engine::region_finish(int i, StepArgs& a, bool blend_deferred) {
    pending_blend = blend_deferred;
    if (sched_kind == RT_SCHED_EULER) a.dsigma = table[i + 1] - table[i];
    else pndm_coeffs(this, i, a);
    launch_step_epilogue(a, stream);
    steps_done++;
}

void rt_engine::region_step(int i, float g, double isa, double ibg, bool xl, bool elide, bool defer_blend) {
    FwdIn in; StepArgs a; bool blend_deferred, inject;
    region_plan(i, g, isa, ibg, xl, elide, defer_blend, in, a, blend_deferred, inject);
    unet_forward(in);
    region_finish(i, a, blend_deferred);
}

// ---- intra-image split of a step over `nparts` GPUs.  The F forwards of a step are independent except that the
// region streams of an injected step consume the self-attention Q / K and one ResNet feature of the text_ref stream, layer by layer
// (0.42 GB per step if shipped).  The streams are therefore cut into `nparts` CONTIGUOUS ranges of the step's stream list
// [uncond, base, uncond_ref, text_ref, region 0 ..] such that text_ref and every region stream land in the same (the last) range: no
// per-layer traffic at all, ONE exchange of the noise predictions (F x 4 x h x w fp32: 1.8 MB at SDXL) per step, after which every
// rank runs the (elementwise) epilogue on the full set and holds identical latents.  With 2 parts: {uncond, base, uncond_ref} |
// {text_ref, regions} while the injection is on, halves otherwise.  Batch invariance (a stream's forward does not depend on the
// other streams of its launch) makes the split run bit-identical with the one-GPU step.
static void region_split_range(int F, int s_tref, bool inject, int part, int nparts, int* first, int* count) {
    RT_REQUIRE(nparts >= 1 && part >= 0 && part < nparts && F >= 1, "split: part index");
    // even prefix split of the stream list; while the injection is on, everything from text_ref on forms the LAST part and the streams
    // before it (uncond, base, uncond_ref: independent forwards) are dealt evenly to the other parts
    int lo, hi;
    if (inject && s_tref >= 0 && nparts > 1) {
        const int head = s_tref, np = nparts - 1;
        auto bound = [&](int k) { return (int)(((long)head * k + np - 1) / np); };       // ceil(head k / (nparts - 1))
        if (part == nparts - 1) { lo = head; hi = F; } else { lo = bound(part); hi = bound(part + 1); }
    } else {
        auto bound = [&](int k) { return (int)(((long)F * k + nparts - 1) / nparts); };   // ceil(F k / nparts)
        lo = bound(part); hi = bound(part + 1);
    }
    *first = lo; *count = hi > lo ? hi - lo : 0;
}

void rt_engine::region_step_part(int i, float g, double isa, double ibg, bool xl, bool elide, bool defer_blend, int part, int nparts, int* first,
                                 int* count, int* plan_info) {
    FwdIn in; StepArgs a; bool blend_deferred, inject;
    region_plan(i, g, isa, ibg, xl, elide, defer_blend, in, a, blend_deferred, inject);
    region_split_range(in.B, a.s_tref, inject, part, nparts, first, count);
    if (plan_info) { plan_info[0] = in.B; plan_info[1] = a.s_tref; plan_info[2] = inject ? 1 : 0; }      // what rt_op_split_range needs for the other parts' ranges
    if (*count == 0) return;
    FwdIn sub{}; sub.h = in.h; sub.w = in.w; sub.t = in.t; sub.B = *count;
    sub.eps_out = eps + (size_t)*first * lat_h * lat_w * 4;          // the range is contiguous: its predictions land in their own slots
    for (int b = 0; b < *count; ++b) {
        const int s = *first + b;
        sub.x[b] = in.x[s]; sub.scale[b] = in.scale[s]; sub.prompt[b] = in.prompt[s]; sub.fontsize[b] = in.fontsize[s];
        RT_REQUIRE(in.qk_src[s] >= *first && in.qk_src[s] < *first + *count, "split: a stream's Q / K source lies outside its part");
        sub.qk_src[b] = in.qk_src[s] - *first;
        sub.res_src[b] = in.res_src[s] < 0 ? -1 : in.res_src[s] - *first;
    }
    unet_forward(sub);
}

void rt_engine::region_step_finish(int i, float g, double isa, double ibg, bool xl, bool elide, bool defer_blend) {
    FwdIn in; StepArgs a; bool blend_deferred, inject;
    region_plan(i, g, isa, ibg, xl, elide, defer_blend, in, a, blend_deferred, inject);
    region_finish(i, a, blend_deferred);
}

// The plain-text step (rd.py:200-214 / xl.py:880-905) in the same three pieces as the rich-text step: the forwards of a contiguous range of
// its two streams [uncond, text], and the epilogue on both.  `first < 0`: both streams (the one-GPU step).
void rt_engine::plain_forward(int i, int first, int count) {
    require_bound();
    const int n = (int)timesteps.size();
    RT_REQUIRE(i >= 0 && i < n, "plain_step: step index out of range");
    RT_REQUIRE(n_prompts >= 2, "plain_step: need [negative, text] prompts");
    RT_REQUIRE(first >= 0 && count >= 1 && first + count <= 2, "plain_step: stream range");
    const bool xl = sched_kind == RT_SCHED_EULER;
    FwdIn in{}; in.h = lat_h; in.w = lat_w; in.t = timesteps[i]; in.B = count;
    in.eps_out = eps + (size_t)first * lat_h * lat_w * 4;            // a stream's prediction lands in its own slot of the eps buffer
    const float scale = xl ? 1.f / std::sqrt(table[i] * table[i] + 1.f) : 1.f;
    for (int b = 0; b < count; ++b) { in.x[b] = lat; in.scale[b] = scale; in.prompt[b] = first + b; in.fontsize[b] = 0; in.qk_src[b] = b; in.res_src[b] = -1; }
    // hooks keep the conditional half: out[1][0][1:2] (rd.py:417,425 / xl.py:980,991) - recorded by the rank that runs the text stream
    if (any_store() && first <= 1 && first + count > 1) in.store_stream = 1 - first;
    unet_forward(in);
}
void rt_engine::plain_finish(int i, float g) {
    const bool xl = sched_kind == RT_SCHED_EULER;
    StepArgs a{};
    a.eps = eps; a.masks = masks; a.lat = lat; a.lat_ref = lat_ref; a.HW = lat_h * lat_w; a.R = 0; a.g = g; a.plain = 1;
    a.s_uncond = 0; a.s_base = 1; a.s_uref = a.s_tref = -1; a.sched = sched_kind; a.step_ref = 0; a.blend = 0;
    if (xl) a.dsigma = table[i + 1] - table[i];
    else pndm_coeffs(this, i, a);
    launch_step_epilogue(a, stream);
    steps_done++;
}
void rt_engine::plain_step(int i, float g) {
    plain_forward(i, 0, 2);
    plain_finish(i, g);
}
// Intra-image split of the plain pass (round 6): part 0 runs the unconditional stream, part 1 the text stream (and records the token maps),
// further parts run nothing; one exchange of the two noise predictions, then rt_plain_step_finish on every rank.
void rt_engine::plain_step_part(int i, int part, int nparts, int* first, int* count) {
    RT_REQUIRE(nparts >= 1 && part >= 0 && part < nparts, "plain_step_part: part index");
    if (nparts == 1) { *first = 0; *count = 2; }
    else { *first = part < 2 ? part : 2; *count = part < 2 ? 1 : 0; }
    if (*count > 0) plain_forward(i, *first, *count);
    else { require_bound(); RT_REQUIRE(i >= 0 && i < (int)timesteps.size(), "plain_step: step index out of range"); }
}
