// translation unit of tests/test_capi_symbols.py::test_amrex_adapter_compiles_against_the_api_mock: every entry point of the adapter is used once
#include <castro_hydro_amd_amrex.H>
void use (amrex::MultiFab& a, amrex::MultiFab& b, amrex::Vector<std::unique_ptr<amrex::MultiFab>>& f, amrex::Vector<std::unique_ptr<amrex::MultiFab>>& m,
          const amrex::Geometry& g, const amrex::BCRec& bc, const castro_amd_params& p)
{
    castro_amd::construct_ctu_hydro_source(a, b, f, m, g, bc, p, 0.0, 1.0);
    castro_amd::construct_ctu_hydro_source_mf(a, b, f, m, g, bc, p, 0.0, 1.0);
    castro_amd::construct_ctu_hydro_source_mf(a, b, f, m, g, bc, p, 0.0, 1.0, 4, &a);
    const amrex::Real grav[3] = { 0.0, 0.0, -1.0 };
    castro_amd::do_sources(0, a, b, a, m, grav, 4, nullptr, g, bc, p, 1.0);
    castro_amd::do_sources(1, a, b, a, m, nullptr, 4, nullptr, g, bc, p, 1.0);
    castro_amd::fill_boundary(a, g, bc);
    castro_amd::expand_state_and_hydro(a, b, f, m, g, bc, p, 0.0, 1.0, 2);
    castro_amd::halo_plans_clear();
}
