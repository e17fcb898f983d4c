// C++ mirror of the reference's optimizer trait, `math::optimizer::State<Observations, EvalState, Model, Error>`
// (reference src/math/optimizer.rs:9-70), as a CRTP base: a solver provides init / step / eval / stop_criterion and
// inherits iterative_solve. Header-only, no dependencies.
#pragma once
#include <cstddef>
#include <optional>
#include <utility>

namespace vors {
namespace optimizer {

// optimizer.rs:9-14
enum class Continue { Stop, Forward };

// Result of iterative_solve: Ok((state, nb_iter)) or Err(error) (optimizer.rs:57).
template <class S, class Error>
struct SolveResult {
    std::optional<S> state;
    std::size_t nb_iter = 0;
    std::optional<Error> error;
    bool ok() const { return state.has_value(); }
};

// Derived must provide:
//   static Derived init(const Observations&, Model);
//   bool step(Model* new_model, Error* err) const;                      // false = Err(err): iterations stop
//   EvalState eval(const Observations&, Model new_model) const;
//   static std::pair<Derived, Continue> stop_criterion(Derived self, std::size_t nb_iter, EvalState eval_state);
template <class Derived, class Observations, class EvalState, class Model, class Error>
struct State {
    // optimizer.rs:57-70
    static SolveResult<Derived, Error> iterative_solve(const Observations& obs, Model initial_model) {
        SolveResult<Derived, Error> out;
        Derived state = Derived::init(obs, std::move(initial_model));
        std::size_t nb_iter = 0;
        for (;;) {
            nb_iter += 1;
            Model new_model;
            Error err;
            if (!state.step(&new_model, &err)) {
                out.error = std::move(err);
                return out;
            }
            EvalState eval_state = state.eval(obs, std::move(new_model));
            auto kept = Derived::stop_criterion(std::move(state), nb_iter, std::move(eval_state));
            state = std::move(kept.first);
            if (kept.second == Continue::Stop) {
                out.state = std::move(state);
                out.nb_iter = nb_iter;
                return out;
            }
        }
    }
};

}  // namespace optimizer
}  // namespace vors
